"""The RCCL-backed multi-GPU entry points of the C-ABI (gpp_comm_*), as far as one GPU allows: a world of one runs the same
calls as N ranks (ncclGetUniqueId, ncclCommInitRank, ncclBroadcast on the library stream; the halo exchange degenerates to the
copy of the tile).  The N > 1 logic of the same partition is covered on the CPU by tests/test_dist_gloo.py."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_comm_world_of_one():
    import torch
    import gridpp_amd as gridpp
    from gridpp_amd import _capi
    lib = _capi.lib()
    gridpp.set_device(0)
    r0, r1 = C.c_int(), C.c_int()
    cover = 0
    for r in range(3):
        _capi.check(lib.gpp_row_tile(10, r, 3, C.byref(r0), C.byref(r1)))
        assert r0.value == cover
        cover = r1.value
    assert cover == 10
    ident = C.create_string_buffer(128)
    _capi.check(lib.gpp_comm_unique_id(ident))
    assert any(b != 0 for b in ident.raw)
    _capi.check(lib.gpp_comm_init(0, 1, ident))
    rank, world = C.c_int(-1), C.c_int(-1)
    _capi.check(lib.gpp_comm_rank(C.byref(rank), C.byref(world)))
    assert (rank.value, world.value) == (0, 1)
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    _capi.check(lib.gpp_comm_broadcast(C.c_void_p(x.data_ptr()), C.c_size_t(4000), 0))
    stream = C.c_void_p()
    _capi.check(lib.gpp_get_stream(C.byref(stream)))
    torch.cuda.synchronize()
    assert float(x.sum()) == 499500.0
    h = np.arange(50, dtype=np.float32)
    _capi.check(lib.gpp_comm_broadcast_host(h.ctypes.data_as(C.c_void_p), C.c_size_t(200), 0))
    assert h.sum() == 1225
    tile = torch.rand((6, 5, 3), device="cuda")
    padded = torch.empty_like(tile)
    top = C.c_int(-1)
    torch.cuda.synchronize()
    _capi.check(lib.gpp_comm_halo_exchange(C.c_void_p(tile.data_ptr()), 6, C.c_size_t(15), 2, C.c_void_p(padded.data_ptr()), C.byref(top)))
    assert top.value == 0 and torch.equal(padded, tile)
    with pytest.raises(ValueError):
        _capi.check(lib.gpp_comm_init(0, 1, ident))        # a communicator exists already
    _capi.check(lib.gpp_comm_destroy())
    _capi.check(lib.gpp_comm_rank(C.byref(rank), C.byref(world)))
    assert (rank.value, world.value) == (0, 1)
