"""Import stub (generator-only): names only; the non-packed nerfacto path never calls them."""


class OccGridEstimator: pass


def accumulate_along_rays(*a, **k): raise NotImplementedError
def pack_info(*a, **k): raise NotImplementedError
def render_weight_from_density(*a, **k): raise NotImplementedError
