"""CPU checks of the library's run-time switches (pt_set_option / pt_get_option): the table in capi.hip, the header's
list and the error paths.  No GPU call is made: the option table is plain host state."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name -> (default, a value inside the range, values that must be rejected)
OPTIONS = {
    "resid_bf16": (1, 0, (-1, 2)),
    "fused_tail": (-1, 1, (-2, 2)),
    "small_rows": (2048, 0, (-1,)),
    "row_tile": (0, 10, (1, 3, 6, 12, -2)),
    "attn_train_rowwise": (0, 1, (-1, 2)),
    "conv8": (1, 0, (-1, 2)),
    "store_overlap": (1, 0, (-1, 2)),
    "row2": (0, 1, (-1, 2)),
    "row2_min_rows": (20480, 1, (-1,)),
    "split_bf16": (1, 0, (-1, 2)),
    "sm_rows": (4800, 0, (-1,)),
    "w_ktile": (1, 0, (-1, 2)),
}


@pytest.fixture(scope="module")
def L():
    import torch  # noqa: F401  (the library links against the HIP runtime torch bundles)
    from probtalk_amd import _lib
    return _lib.lib()


def _get(L, name):
    v = C.c_int(-12345)
    assert L.pt_get_option(name.encode(), C.byref(v)) == 0, L.pt_last_error()
    return v.value


def test_every_option_has_its_default_takes_a_value_and_rejects_out_of_range(L):
    for name, (default, ok, bad) in OPTIONS.items():
        assert _get(L, name) == default, name
        try:
            assert L.pt_set_option(name.encode(), ok) == 0, (name, L.pt_last_error())
            assert _get(L, name) == ok
            for b in bad:
                assert L.pt_set_option(name.encode(), b) != 0, f"{name} = {b} was accepted"
                assert name.encode() in L.pt_last_error()
                assert _get(L, name) == ok, "a rejected value must leave the option unchanged"
        finally:
            assert L.pt_set_option(name.encode(), default) == 0


def test_unknown_option_names_every_option_in_the_error(L):
    assert L.pt_set_option(b"no_such_switch", 1) != 0
    msg = L.pt_last_error().decode()
    for name in OPTIONS:
        assert name in msg, f"{name} missing from: {msg}"
    assert L.pt_set_option(None, 1) != 0
    v = C.c_int()
    assert L.pt_get_option(b"no_such_switch", C.byref(v)) != 0
    assert L.pt_get_option(b"row2", None) != 0


def test_header_documents_every_option():
    txt = open(os.path.join(ROOT, "include", "probtalk_hip.h")).read()
    doc = txt[txt.index("The library's only run-time switches"):]
    listed = set(re.findall(r'^ \*   "([a-z0-9_]+)"', doc, flags=re.M))
    assert listed == set(OPTIONS), sorted(listed ^ set(OPTIONS))
