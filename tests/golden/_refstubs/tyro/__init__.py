"""Import stub (generator-only): nerfstudio.cameras.camera_optimizers annotates two config fields with
tyro.conf.Suppress[...]; nothing of tyro is called on the paths the fixture generators use."""


class _Sub:
    def __class_getitem__(cls, item):
        return item
class conf:
    Suppress = _Sub
    FlagConversionOff = _Sub
    Fixed = _Sub
    @staticmethod
    def subcommand(*a, **k):
        return None
def cli(*a, **k):
    raise NotImplementedError
