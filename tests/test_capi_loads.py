"""The C-ABI library loads and exports every symbol the public headers declare (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from ray_tracing_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in ("rt_amd.h", "rt_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _capi.load()
    names = declared_functions()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"
        assert n in _capi.PROTOTYPES, f"{n} has no ctypes prototype"


def test_abi_version_and_struct_sizes():
    lib = _capi.load()
    assert lib.rt_abi_version() == 1
    assert C.sizeof(_capi.rt_scene_desc) == lib.rt_host_scene_desc.restype._type_.__dict__.get("_size_", C.sizeof(_capi.rt_scene_desc))


def test_host_layer_errors_are_reported():
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.rt_host_scene_build(b"no_such_scene", C.c_float(1.0), C.byref(h)) == _capi.RT_ERR_INVALID
    assert b"no_such_scene" in lib.rt_last_error()
    assert lib.rt_host_scene_build(b"cornell_box", C.c_float(-1.0), C.byref(h)) == _capi.RT_ERR_INVALID
    assert "cornell_box" in lib.rt_host_scene_names().decode()


def test_scene_description_is_validated_before_any_device_work(cornell_host):
    """rt_scene_create rejects corrupt tables with RT_ERR_INVALID (checked before touching HIP)."""
    lib = _capi.load()
    import copy
    d = _capi.rt_scene_desc.from_buffer_copy(cornell_host.desc.contents)
    out = C.c_void_p()
    d.abi_version = 99
    assert lib.rt_scene_create(C.byref(d), 0, C.byref(out)) == _capi.RT_ERR_INVALID
    d = _capi.rt_scene_desc.from_buffer_copy(cornell_host.desc.contents)
    entries = (_capi.rt_entry * d.n_entries)(*[d.entries[i] for i in range(d.n_entries)])
    entries[0].a = 0  # a skip link that does not move forward would hang the scan
    d.entries = entries
    assert lib.rt_scene_create(C.byref(d), 0, C.byref(out)) == _capi.RT_ERR_INVALID
    assert b"skip link" in lib.rt_last_error()
    d = _capi.rt_scene_desc.from_buffer_copy(cornell_host.desc.contents)
    d.n_lights = 0
    assert lib.rt_scene_create(C.byref(d), 0, C.byref(out)) == _capi.RT_ERR_INVALID


def test_descriptions_on_which_the_render_paths_would_disagree_are_rejected(cornell_host):
    """A light under an instance chain (translate / rotate_y do not forward pdf_value / random, src/hitable.h:19-20: the
    flattener passes -1 for it) and a list member whose chain differs from its list's: the exact, record-table and per-lane
    paths would each read such a description differently, so rt_scene_create refuses it."""
    lib = _capi.load()
    out = C.c_void_p()
    src = cornell_host.desc.contents

    def copy_with_prims():
        d = _capi.rt_scene_desc.from_buffer_copy(src)
        prims = (_capi.rt_prim * d.n_prims)(*[d.prims[i] for i in range(d.n_prims)])
        d.prims = prims
        return d, prims

    assert src.n_chains >= 2  # the tall box sits under translate(rotate_y(...))
    d, prims = copy_with_prims()
    light = [src.lights[i] for i in range(src.n_lights) if src.lights[i] >= 0][0]
    prims[light].chain = 1
    assert lib.rt_scene_create(C.byref(d), 0, C.byref(out)) == _capi.RT_ERR_INVALID
    assert b"instance chain" in lib.rt_last_error()

    d, prims = copy_with_prims()
    group = [i for i in range(src.n_prims) if src.prims[i].kind == _capi.RT_PRIM_GROUP and src.prims[i].chain != 0][0]
    prims[prims[group].first + 2].chain = 0
    assert lib.rt_scene_create(C.byref(d), 0, C.byref(out)) == _capi.RT_ERR_INVALID
    assert b"its group chain" in lib.rt_last_error()

    d = _capi.rt_scene_desc.from_buffer_copy(src)
    lights = (C.c_int32 * d.n_lights)(*([-2] * d.n_lights))
    d.lights = lights
    assert lib.rt_scene_create(C.byref(d), 0, C.byref(out)) == _capi.RT_ERR_INVALID
