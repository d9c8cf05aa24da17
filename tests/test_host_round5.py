"""Host-side pieces of round 5 that the GPU kernels rely on (CPU tier)."""
import numpy as np

from peppa_pig_face_landmark_amd.graph import ir


def test_presplit_weights_match_the_kernels_old_per_wave_split():
    """ir.py::_f32_or_presplit packs, for split programs, the 16 bytes of every four f32 weights as [hi x 4 | lo x 4] f16 with
    x = w / unscale (a power of two), hi = f16(x), lo = f16(x - f32(hi)): exactly what mbconv_wave_f32_kernel computed in every wave
    before round 5 (k_mbconv.h).  f32 programs keep plain f32."""
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((48, 32)) * 0.37).astype(np.float32)
    for dtype, split in (("f32s", True), ("f32", False)):
        pb = ir.ProgramBuilder(dtype, 64, 64)
        unscale = pb._pow2_unscale(w)
        off = pb._f32_or_presplit(w, unscale)
        raw = bytes(pb.consts[off:off + w.size * 4])
        if not split:
            assert unscale == 1.0
            assert np.array_equal(np.frombuffer(raw, np.float32).reshape(w.shape), w)
            continue
        assert np.log2(unscale) == np.round(np.log2(unscale))                     # a power of two
        packed = np.frombuffer(raw, np.float16).reshape(w.shape[0], w.shape[1] // 4, 8)
        x = w * np.float32(1.0 / unscale)
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)
        assert np.array_equal(packed[:, :, :4].reshape(w.shape), hi)
        assert np.array_equal(packed[:, :, 4:].reshape(w.shape), lo)
        # the scale keeps the low halves out of the f16 subnormal range for the largest weights, and hi + lo carries ~22 bits
        rec = (hi.astype(np.float64) + lo.astype(np.float64)) * unscale
        assert np.abs(rec - w).max() <= np.abs(w).max() * 2.0 ** -21
        assert 8192.0 <= np.abs(x).max() <= 16384.0


def test_magic_division_is_exact_on_its_documented_domain():
    """pf_common.h pf_div_small: (x * m) >> 20 with m = 2^20 / d + 1 equals x / d for 0 <= x < min(4096, 2^20 / d) -- the per-lane tile
    arithmetic of the stem / detector / crop kernels stays inside that range (region pixels < 512, staging words < 1024, crop
    rows x 256 < 2048 with d <= 256)."""
    for d in list(range(1, 300)) + [304, 400, 480, 512, 1024]:
        m = (1 << 20) // d + 1
        x = np.arange(0, min(4096, (1 << 20) // d), dtype=np.uint64)
        assert (x * m < 2 ** 32).all()
        assert np.array_equal((x * m) >> 20, x // d), d
