"""Run the REFERENCE'S OWN unit-test modules, unmodified, against this package's mirror classes (CPU; the kernel wrappers
replaced by the oracle's restatements, tests/cpu_kernels.py). Own process: `nerfstudio.<module>` is aliased in sys.modules
(and on the parent package) to a proxy that answers with this package's class where one exists and with the reference's own
symbol otherwise (classes of other methods the test file also imports).

    python tests/refunit/run.py <file relative to /root/reference/tests> ...   ->  one JSON line
    {"<file>::<test>": ["pass"] | ["fail", "<ExceptionType>", "<message>"], ..., "<file>": {"mirrored": [class names]}}
"""
import importlib
import importlib.util
import inspect
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refdrive  # noqa: E402

refdrive.install()
import cpu_kernels  # noqa: E402

MIRRORED = ["cameras.rays", "field_components.field_heads", "field_components.encodings", "field_components.mlp",
            "field_components.embedding", "field_components.spatial_distortions", "fields.nerfacto_field", "fields.density_fields",
            "fields.base_field", "model_components.losses", "model_components.ray_samplers", "model_components.renderers",
            "model_components.scene_colliders", "utils.tensor_dataclass"]


class Proxy(types.ModuleType):
    def __init__(self, name, ours, ref):
        super().__init__(name)
        self.__dict__["_ours"], self.__dict__["_ref"] = ours, ref

    def __getattr__(self, key):
        return getattr(self._ours, key) if hasattr(self._ours, key) else getattr(self._ref, key)


class _Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def main(files):
    for a in MIRRORED:
        ref = importlib.import_module("nerfstudio." + a)
        ours = importlib.import_module("nerfstudio_amd." + a)
        px = Proxy("nerfstudio." + a, ours, ref)
        sys.modules["nerfstudio." + a] = px
        parent, leaf = ("nerfstudio." + a).rsplit(".", 1)
        setattr(sys.modules[parent], leaf, px)  # `from nerfstudio.field_components import encodings`
    cpu_kernels.installed(_Patch()).__enter__()
    out = {}
    for f in files:
        spec = importlib.util.spec_from_file_location("refunit_" + f.replace("/", "_")[:-3], os.path.join(refdrive.REF, "tests", f))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out[f] = {"mirrored": sorted({v.__name__ for v in vars(mod).values() if getattr(v, "__module__", "").startswith("nerfstudio_amd")})}
        for name in sorted(n for n in dir(mod) if n.startswith("test_") and callable(getattr(mod, n))):
            fn = getattr(mod, name)
            if inspect.signature(fn).parameters:
                continue  # (fixtures / parametrised: none among the files used)
            try:
                fn()
                out[f"{f}::{name}"] = ["pass"]
            except Exception as e:  # noqa: BLE001
                out[f"{f}::{name}"] = ["fail", type(e).__name__, str(e)[:300]]
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1:])
