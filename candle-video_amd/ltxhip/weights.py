"""Weight ingestion over include/ltxhip_weights.h: thin ctypes mirror of the reference's
`weight_format.rs` (WeightFormat, KeyRemapper) and `loader.rs` (WeightLoader name mapping, SafetensorsIndex, directory
resolution, validate_tensor_names).  All logic lives in the C++ library; this file only marshals strings."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

from . import lib, _check

DIFFUSERS, OFFICIAL = 0, 1            # WeightFormat (weight_format.rs:13-19)
_CAP = 4096


def detect_format(path: str) -> int:
    return int(lib.ltx_weights_detect_format(path.encode()))


def remap_key(key: str) -> str:
    """KeyRemapper::remap_key (weight_format.rs:55-83)."""
    buf = C.create_string_buffer(_CAP)
    _check(lib.ltx_weights_remap_key(key.encode(), buf, _CAP))
    return buf.value.decode()


def is_transformer_key(key: str) -> bool:
    return bool(lib.ltx_weights_is_transformer_key(key.encode()))


def is_vae_key(key: str) -> bool:
    return bool(lib.ltx_weights_is_vae_key(key.encode()))


class WeightLoader:
    """Name-mapping part of loader.rs's WeightLoader (:201-317): ordered exact / prefix / suffix rules."""

    def __init__(self):
        self._h = C.c_void_p(lib.ltx_name_mapper_create())

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ltx_name_mapper_destroy(self._h)
            self._h = None

    def _add(self, kind: int, a: str, b: str) -> "WeightLoader":
        _check(lib.ltx_name_mapper_add(self._h, kind, a.encode(), b.encode()))
        return self

    def add_mapping(self, frm: str, to: str) -> "WeightLoader":
        return self._add(0, frm, to)

    def add_prefix_mapping(self, frm: str, to: str) -> "WeightLoader":
        return self._add(1, frm, to)

    def add_suffix_mapping(self, frm: str, to: str) -> "WeightLoader":
        return self._add(2, frm, to)

    def has_mapping(self, name: str) -> bool:
        return bool(lib.ltx_name_mapper_has_mapping(self._h, name.encode()))

    def map_name(self, name: str) -> str:
        buf = C.create_string_buffer(_CAP)
        _check(lib.ltx_name_mapper_map(self._h, name.encode(), buf, _CAP))
        return buf.value.decode()


def validate_tensor_names(expected: Sequence[str], actual: Sequence[str]) -> List[str]:
    """validate_tensor_names (loader.rs:495-505): expected names missing from actual, in order."""
    e = (C.c_char_p * max(len(expected), 1))(*[x.encode() for x in expected])
    a = (C.c_char_p * max(len(actual), 1))(*[x.encode() for x in actual])
    idx = (C.c_size_t * max(len(expected), 1))()
    n = C.c_size_t(0)
    _check(lib.ltx_weights_validate_names(e, len(expected), a, len(actual), idx, C.byref(n)))
    return [expected[idx[i]] for i in range(n.value)]


def resolve_weight_files(path: str) -> List[str]:
    """File resolution of WeightLoader::load_from_directory (loader.rs:341-397) or a single file as is."""
    cap = 1 << 16
    buf = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    _check(lib.ltx_weights_resolve(path.encode(), buf, cap, C.byref(n)))
    return [p.decode() for p in buf.raw.split(b"\0")[: n.value]]


class SafetensorsFile:
    """One mmap'ed safetensors file: names, dtype strings, shapes and raw payload views (no host copies)."""

    def __init__(self, path: str):
        self._h = C.c_void_p()
        _check(lib.ltx_safetensors_open(path.encode(), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib.ltx_safetensors_close(self._h)
            self._h = None

    __del__ = close

    def __len__(self) -> int:
        return int(lib.ltx_safetensors_count(self._h))

    def tensor(self, i: int) -> Tuple[str, str, Tuple[int, ...], bytes]:
        name, dt = C.c_char_p(), C.c_char_p()
        nd, shp, data, nb = C.c_int(), C.POINTER(C.c_int64)(), C.c_void_p(), C.c_size_t()
        _check(lib.ltx_safetensors_tensor(self._h, i, C.byref(name), C.byref(dt), C.byref(nd), C.byref(shp), C.byref(data), C.byref(nb)))
        return name.value.decode(), dt.value.decode(), tuple(shp[k] for k in range(nd.value)), C.string_at(data.value, nb.value) if nb.value else b""

    def tensors(self) -> Dict[str, Tuple[str, Tuple[int, ...], bytes]]:
        return {n: (d, s, b) for n, d, s, b in (self.tensor(i) for i in range(len(self)))}
