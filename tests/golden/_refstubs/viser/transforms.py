"""Import stub (generator-only)."""


class SO3: pass
class SE3: pass
