"""Stand-in for dm-tree: map_structure over nested dict/list/tuple (residue_constants.py:24,1082)."""


def map_structure(fn, *structs):
    s0 = structs[0]
    if isinstance(s0, dict):
        return {k: map_structure(fn, *[s[k] for s in structs]) for k in s0}
    if isinstance(s0, (list, tuple)):
        out = [map_structure(fn, *xs) for xs in zip(*structs)]
        return type(s0)(out) if not hasattr(s0, "_fields") else type(s0)(*out)
    return fn(*structs)
