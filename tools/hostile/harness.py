"""What the hostile soaks share (tools/oi_hostile_soak.py, ensi_hostile_soak.py, ensi_multi_hostile_soak.py, nbh_hostile_soak.py).

`poison` on the command line selects the -DGPP_POISON build of the library (gridpp_amd/lib/var_poison.so, tools/hostile/build.sh; every
fresh allocation of that build is filled with 0xFF) and makes `Hostile.before_call()` fill, before EVERY call into the library,
  * all 160 KB of LDS of every CU and 500 VGPRs / AGPRs per lane of every SIMD with 0xFF (tools/hostile/poison.hip), and
  * every byte of every call-to-call workspace of the library with 0xFF (gpp_debug_poison_workspaces: OI, EnSI + ensi_multi,
    neighbourhood -- lists, counters, parks, Gram matrices, parked selections, byte planes, row sums),
so that a kernel that reads what THIS call never wrote meets NaNs, negative indices and absurd counts instead of the remains of
the previous call.  Must be imported BEFORE gridpp_amd (it sets GPP_LIB)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class Hostile:
    def __init__(self, poison):
        self.poison = poison
        if poison:
            os.environ["GPP_LIB"] = os.path.join(ROOT, "gridpp_amd", "lib", "var_poison.so")
        import gridpp_amd
        self.gridpp = gridpp_amd
        self.glib = gridpp_amd._capi.lib()       # (first: it loads torch's HIP runtime, which the helper library then shares)
        self.plib = None
        if poison:
            self.plib = C.CDLL(os.path.join(ROOT, "tools", "hostile", "libpoison.so"))
            self.glib.gpp_debug_poison_workspaces.argtypes = [C.c_int, C.c_int]
            self.glib.gpp_debug_poison_workspaces.restype = C.c_int
        self.calls = 0

    def before_call(self, keep_padding=0):
        """keep_padding = 1: the byte planes of the fused quantile_fast path keep the padding of the remembered layout (the cache of
        neighbourhood.hip is exercised); every cell byte is poisoned all the same"""
        self.calls += 1
        if not self.poison:
            return
        assert self.plib.poison_lds(C.c_uint(0xFFFFFFFF)) == 0
        assert self.plib.poison_regs(C.c_uint(0xFFFFFFFF)) == 0
        assert self.glib.gpp_debug_poison_workspaces(0xFF, int(keep_padding)) == 0


CACHE = os.path.join(ROOT, "build", "hostile_cache")     # (build/ is git-ignored and travels to the GPU box with the snapshot)


def cached(tool, seed, inputs, compute):
    """the oracle's answer for (tool, seed): computed once -- e.g. in the build container with `... LO HI 0` -- and kept on disk, keyed on
    a digest of the inputs so that a changed generator never meets a stale answer; `compute()` returns a dict of arrays"""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(inputs):
        v = inputs[k]
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes() if isinstance(v, np.ndarray) else repr(v).encode())
    path = os.path.join(CACHE, "%s_%d_%s.npz" % (tool, seed, h.hexdigest()[:16]))
    if os.path.exists(path):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    out = compute()
    os.makedirs(CACHE, exist_ok=True)
    tmp = path + ".tmp%d.npz" % os.getpid()
    np.savez(tmp, **out)
    os.replace(tmp, path)
    return out


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def plain_mismatch(out, ref, floor, rtol=1e-5):
    """None when |out - ref| / max(|ref|, floor) < rtol everywhere and the NaN / inf patterns agree; otherwise what differs"""
    out, ref = np.asarray(out), np.asarray(ref)
    if out.shape != ref.shape:
        return "shape %s vs %s" % (out.shape, ref.shape)
    dn = np.isnan(out) != np.isnan(ref)
    if dn.any():
        return "NaN pattern differs in %d values, first at %s" % (int(dn.sum()), np.argwhere(dn)[0].tolist())
    inf = np.isinf(ref)
    if inf.any() and not (out[inf] == ref[inf]).all():
        return "infinities differ"
    m = ~np.isnan(ref) & ~inf
    if m.any():
        err = np.abs(out[m].astype(np.float64) - ref[m]) / np.maximum(np.abs(ref[m]), floor)
        if not err.max() < rtol:
            return "max rel err %.3g (%d values at or above %.0e)" % (err.max(), int((err >= rtol).sum()), rtol)
    return None
