"""Test infrastructure (authoring container only): import the REFERENCE's trainer-side modules — engine/trainer.py,
pipelines/base_pipeline.py, engine/optimizers.py, models/nerfacto.py — read-only from /root/reference, although most of
their third-party imports (tyro, viser, torchvision, imageio, cv2, rich extras, ...) are not installed here.

`install()` puts tests/golden/_refstubs and /root/reference on sys.path and registers a meta-path finder that answers
every import of an ABSENT third-party package with a permissive placeholder module (any attribute is another placeholder,
callable, subscriptable, usable as a base class or in a type union). Nothing of those packages is executed by the code
the tests drive: `Trainer.train_iteration` (engine/trainer.py:487-531), `VanillaPipeline.get_train_loss_dict`
(pipelines/base_pipeline.py:290-303), `Optimizers` (engine/optimizers.py:74-193) and the model classes are plain torch.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
STUBS = os.path.join(os.path.dirname(HERE), "golden", "_refstubs")
_PERMISSIVE_ROOTS = {"cv2", "tyro", "viser", "torchvision", "imageio", "appdirs", "newrawpy", "rawpy", "pyquaternion", "mediapy",
                     "open3d", "trimesh", "pymeshlab", "xatlas", "plotly", "nuscenes", "comet_ml", "wandb", "splines", "h5py",
                     "msgpack_numpy", "gsplat", "tinycudann", "pytorch_msssim", "torchmetrics", "tensorboard", "gdown", "skimage",
                     "pycolmap", "hloc", "lpips", "fpsample", "timm", "nerfacc", "tensorly", "jaxtyping_placeholder", "matplotlib", "PIL", "scipy_placeholder"}


class Placeholder(types.ModuleType):
    """A module / class / function / constant all at once."""

    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        child = Placeholder(f"{self.__name__}.{name}")
        object.__setattr__(self, name, child)
        return child

    def __call__(self, *args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]  # used as a decorator
        return self

    def __mro_entries__(self, bases):
        return (object,)

    def __getitem__(self, item):
        return self

    def __or__(self, other):
        return self

    __ror__ = __or__

    def __iter__(self):
        return iter(())

    def __bool__(self):
        return False


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] not in _PERMISSIVE_ROOTS:
            return None
        for finder in sys.meta_path:  # a real installation wins
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            try:
                spec = finder.find_spec(name, path, target)
            except Exception:  # noqa: BLE001
                spec = None
            if spec is not None and STUBS not in (spec.origin or ""):
                return None
        return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return Placeholder(spec.name)

    def exec_module(self, module):
        pass


_installed = False


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "nerfstudio"))


def install() -> None:
    global _installed
    if _installed:
        return
    for name in list(sys.modules):  # stub modules imported earlier (fixture generators' minimal stubs) give way
        if name.split(".")[0] in ("cv2", "tyro", "viser") and STUBS in (getattr(sys.modules[name], "__file__", "") or ""):
            del sys.modules[name]
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, STUBS)  # jaxtyping (real subscriptable stand-ins), nerfacc names
    sys.path.insert(1, REF)
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    _installed = True
