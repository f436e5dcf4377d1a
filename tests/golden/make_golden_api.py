"""api_snapshot: the public-signature snapshot shared by make_golden.py (applied to the reference) and
tests/test_api_surface.py (applied to the engine).  Imports neither."""


def api_snapshot(mod):
    """{qualified name: [[parameter, kind, repr(default) | None], ...]} of the public surface SURVEY 8(b) lists: the
    three entry points, every public method of the two classes, the module-level helpers.  Shared with
    tests/test_api_surface.py, which applies it to the engine."""
    import inspect

    def sig(fn):
        return [[p.name, p.kind.name, None if p.default is p.empty else repr(p.default)]
                for p in inspect.signature(fn).parameters.values()]
    snap = {}
    for cname in ("NoiseScheduleVP", "DPM_Solver"):
        cls = getattr(mod, cname)
        for name, member in inspect.getmembers(cls, predicate=inspect.isfunction):
            if name == "__init__" or not name.startswith("_"):
                snap["%s.%s" % (cname, name)] = sig(member)
    for fname in ("model_wrapper", "interpolate_fn", "expand_dims"):
        snap[fname] = sig(getattr(mod, fname))
    return snap
