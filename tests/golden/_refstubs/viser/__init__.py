"""Import stub (generator-only)."""


class ViserServer: pass
