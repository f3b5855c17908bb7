"""Import stub (generator-only): lets /root/reference import without jaxtyping."""


class _T:
    def __class_getitem__(cls, item):
        return cls


class Float(_T): pass
class Int(_T): pass
class Shaped(_T): pass
class Bool(_T): pass
class UInt8(_T): pass
class Num(_T): pass
