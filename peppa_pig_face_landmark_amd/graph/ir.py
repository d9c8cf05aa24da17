"""Program assembler: builds the packed network program executed by ``csrc/engine.cpp``.

The binary layout is defined in ``csrc/pf_program.h`` -- keep both in sync.  numpy only (the
engine side of the boundary never sees torch / onnxruntime).

A program is a straight-line list of fused layer ops over NHWC activation tensors.  BatchNorm is
folded into the preceding conv here (float64), weights are re-laid-out for the kernels
(``[Npad][taps][Cpad]`` rows of 64-byte K steps for the MFMA implicit-GEMM, ``[taps][C]`` for the
depthwise kernel, ``[K][N]`` for the tiny pooled-vector FCs) and activation buffers are placed in
one arena with lifetime-based reuse.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

MAGIC = 0x47504650
VERSION = 11
OP_FIELDS = 39

DTYPE_F16, DTYPE_F32, DTYPE_F32_SPLIT = 0, 1, 2
ELEM_ACT, ELEM_F32, ELEM_I32, ELEM_U8 = 0, 1, 2, 3
ACT = {"none": 0, "relu": 1, "hswish": 2, "silu": 3, "sigmoid": 4, "hsigmoid": 5}

OP_STEM, OP_CONV, OP_DW, OP_UPCAT, OP_GAP, OP_FC, OP_SCSE, OP_HMDEC, OP_MAXPOOL, OP_COPY, OP_DETDEC, OP_SEPUP, OP_ADDUP, OP_MBCONV, OP_EXPDW, OP_CHAIN, OP_BLOCK, OP_DETUNIT, OP_DETC3, OP_DETSTEM, OP_LMFRONT, OP_HRB, OP_FUSEUP = range(1, 24)
OP_MBX = 24
OP_FC2 = 25
OP_FRONT2 = 26

# conv_gemm_kernel tile configurations (BM, BN, WARPS_M); index == cfg field
CONV_CFGS = [(128, 128, 2), (128, 64, 2), (256, 32, 4), (256, 16, 4)]


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def fold_bn(weight: np.ndarray, conv_bias: Optional[np.ndarray], bn: Optional[Dict[str, np.ndarray]],
            eps: float = 1e-5) -> Tuple[np.ndarray, np.ndarray]:
    """conv (+bias) followed by eval-mode BatchNorm -> (weight', bias') in float64."""
    w = weight.astype(np.float64)
    cout = w.shape[0]
    b = np.zeros(cout, np.float64) if conv_bias is None else conv_bias.astype(np.float64)
    if bn is not None:
        s = bn["weight"].astype(np.float64) / np.sqrt(bn["running_var"].astype(np.float64) + eps)
        w = w * s.reshape((-1,) + (1,) * (w.ndim - 1))
        b = (b - bn["running_mean"].astype(np.float64)) * s + bn["bias"].astype(np.float64)
    return w, b


def bn_affine(bn: Dict[str, np.ndarray], eps: float = 1e-5) -> Tuple[np.ndarray, np.ndarray]:
    """eval-mode BatchNorm as y = s*x + t (float64)."""
    s = bn["weight"].astype(np.float64) / np.sqrt(bn["running_var"].astype(np.float64) + eps)
    t = bn["bias"].astype(np.float64) - bn["running_mean"].astype(np.float64) * s
    return s, t


def _upsample_tap_matrix(y: int, H: int, lo_h: int) -> np.ndarray:
    """A[k][j]: weight of low-res patch row j (rows (y>>1)-1+j, clamped) in the bilinear x2 upsampled
    row y-1+k (torch align_corners=False, source index clamped at 0); all-zero row if y-1+k is padding."""
    A = np.zeros((3, 3), np.float64)
    m = y >> 1
    for k in range(3):
        yy = y - 1 + k
        if yy < 0 or yy >= H:
            continue
        sy = max((yy + 0.5) * 0.5 - 0.5, 0.0)
        y0 = int(sy)
        y1 = y0 + (1 if y0 < lo_h - 1 else 0)
        ly = sy - y0
        A[k][y0 - m + 1] += 1.0 - ly
        A[k][y1 - m + 1] += ly
    return A


class _Buf:
    __slots__ = ("etype", "elems", "first", "last", "pinned", "offset_units", "name")


class _Tensor:
    # C = stored channels (multiple of the 16-byte vector); real_c = meaningful channels (the rest are zeros)
    __slots__ = ("buf", "coff", "ld", "H", "W", "C", "name", "real_c")


class ProgramBuilder:
    def __init__(self, dtype: str, in_h: int, in_w: int, keep_all: bool = False):
        assert dtype in ("f16", "f32", "f32s")
        self.dtype = {"f16": DTYPE_F16, "f32": DTYPE_F32, "f32s": DTYPE_F32_SPLIT}[dtype]
        self.split = dtype == "f32s"         # f32 tensors, split-precision (3 x f16 MFMA) dense convs
        self.np_act = np.float16 if dtype == "f16" else np.float32
        self.esize = 2 if dtype == "f16" else 4
        self.ve = 16 // self.esize          # elements per 16-byte vector
        self.ke = 64 // self.esize          # elements per 64-byte K step of the direct GEMM
        self.in_h, self.in_w = in_h, in_w
        self.keep_all = keep_all
        self.bufs: List[_Buf] = []
        self.tensors: List[_Tensor] = []
        self.ops: List[Tuple[int, List[int], List[int], List[int]]] = []  # code, fields, reads(bufs), writes(bufs)
        self.consts = bytearray()
        self.tensor_names: Dict[str, int] = {}

    # ---- storage --------------------------------------------------------------------------
    def buffer(self, elems: int, etype: int = ELEM_ACT, name: str = "", pinned: bool = False) -> int:
        b = _Buf()
        b.etype, b.elems, b.first, b.last, b.pinned, b.offset_units, b.name = etype, int(elems), None, None, pinned, 0, name
        self.bufs.append(b)
        return len(self.bufs) - 1

    def tensor(self, H: int, W: int, C: int, buf: Optional[int] = None, coff: int = 0, ld: Optional[int] = None,
               name: str = "") -> int:
        assert C % self.ve == 0, f"channel count {C} must be a multiple of {self.ve}"
        if buf is None:
            ld = C if ld is None else ld
            buf = self.buffer(H * W * ld, ELEM_ACT, name)
        assert ld is not None and coff % self.ve == 0 and ld % self.ve == 0
        t = _Tensor()
        t.buf, t.coff, t.ld, t.H, t.W, t.C, t.name, t.real_c = buf, coff, ld, H, W, C, name, C
        self.tensors.append(t)
        tid = len(self.tensors) - 1
        if name:
            self.tensor_names[name] = tid
        return tid

    def view(self, base_buf: int, H: int, W: int, C: int, coff: int, ld: int, name: str = "") -> int:
        return self.tensor(H, W, C, base_buf, coff, ld, name)

    def strided_view(self, base_buf: int, H: int, W: int, C: int, coff: int, ld: int) -> int:
        """Write-only target for channel-interleaved stores (channel n lands at coff + n*cs): no
        vector-alignment requirement because those stores are scalar."""
        t = _Tensor()
        t.buf, t.coff, t.ld, t.H, t.W, t.C, t.name, t.real_c = base_buf, coff, ld, H, W, C, "", C
        self.tensors.append(t)
        return len(self.tensors) - 1

    def const(self, arr: np.ndarray) -> int:
        while len(self.consts) % 256:
            self.consts.append(0)
        off = len(self.consts)
        self.consts += np.ascontiguousarray(arr).tobytes()
        return off

    def const_f32(self, arr) -> int:
        return self.const(np.asarray(arr, np.float64).astype(np.float32))

    def const_act(self, arr) -> int:
        return self.const(np.asarray(arr, np.float64).astype(self.np_act))

    def _op(self, code: int, fields: Sequence[int], reads: Sequence[int], writes: Sequence[int]):
        f = [int(v) for v in fields]
        assert len(f) <= OP_FIELDS
        f += [0] * (OP_FIELDS - len(f))
        self.ops.append((code, f, [r for r in reads if r is not None and r >= 0],
                         [w for w in writes if w is not None and w >= 0]))

    def _tb(self, t: int) -> int:
        return self.tensors[t].buf if t is not None and t >= 0 else -1

    # ---- ops --------------------------------------------------------------------------------
    def stem(self, weight: np.ndarray, bias: np.ndarray, act: str, out_name: str = "") -> int:
        """3x3 stride-2 pad-1 conv on the 3-channel program input; weight [CO,3,3,3] (BN folded), CO % 16 == 0."""
        co = weight.shape[0]
        assert weight.shape == (co, 3, 3, 3) and co % 16 == 0
        oh, ow = (self.in_h + 1) // 2, (self.in_w + 1) // 2
        out = self.tensor(oh, ow, co, name=out_name)
        w = np.transpose(weight.astype(np.float64), (2, 3, 1, 0)).reshape(27, co)  # [(ky,kx,ci)][co]
        off_u8 = self.const_f32(w / 255.0)
        off_f32 = self.const_f32(w)
        off_b = self.const_f32(bias)
        f = [-1, out, off_u8, off_b, ACT[act], off_f32, -1, -1, 0, 0]
        if self.split and co in (16, 64) and self.in_w % 4 == 0 and self.in_h % 2 == 0:
            # f32s programs: the same conv as a split-precision MFMA GEMM on the staged image (csrc/k_front.h stem_mfma_kernel)
            ws = self._stem_k_order(weight)
            wu, su = self._split_rows(ws / 255.0)
            wf, sf = self._split_rows(ws)
            fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]
            f[6:10] = [self.const(wu), self.const(wf), fbits(su), fbits(sf)]
        self._op(OP_STEM, f, [], [self._tb(out)])
        return out

    SPLIT_MIN_CIN = 64       # pointwise convs below this are bandwidth-bound: the exact-f32 direct kernel is as fast
    SPLIT_MIN_CIN_KXK = 16   # kxk convs are matrix-core bound much earlier (HRNet's 18/36-channel 3x3 stacks)

    def conv_uses_split(self, cin: int, taps: int = 1) -> bool:
        return self.split and (cin >= self.SPLIT_MIN_CIN or (taps > 1 and cin >= self.SPLIT_MIN_CIN_KXK))

    def pack_conv_weight(self, weight: np.ndarray, force_split: bool = False) -> Tuple[int, int, int, float, bool]:
        """[N,Cin,KH,KW] -> (const offset, Npad, Cpad, acc_scale, use_split).
        direct kernels: [Npad][KH*KW][Cpad] in the activation dtype (64-byte K steps), acc_scale 1;
        split kernels : [Npad][KH*KW][Cpad/32][hi 32 x f16 | lo 32 x f16] of w * 2^s, acc_scale 2^-s."""
        n, cin, kh, kw = weight.shape
        use_split = force_split or self.conv_uses_split(cin, kh * kw)
        ke = 32 if use_split else 64 // self.esize
        npad, cpad = _round_up(n, 16), _round_up(cin, ke)
        w = np.zeros((npad, kh * kw, cpad), np.float64)
        w[:n, :, :cin] = np.transpose(weight.astype(np.float64), (0, 2, 3, 1)).reshape(n, kh * kw, cin)
        if not use_split:
            return self.const_act(w), npad, cpad, 1.0, False
        wmax = float(np.abs(w).max())
        s = 0 if wmax == 0.0 else int(np.floor(np.log2(16384.0 / wmax)))   # max |w * 2^s| in [2^13, 2^14]
        ws = (w * (2.0 ** s)).astype(np.float32)
        hi = ws.astype(np.float16)
        lo = (ws - hi.astype(np.float32)).astype(np.float16)
        blocks = np.stack([hi.reshape(npad, kh * kw, cpad // 32, 32), lo.reshape(npad, kh * kw, cpad // 32, 32)], axis=3)
        return self.const(np.ascontiguousarray(blocks)), npad, cpad, float(2.0 ** (-s)), True

    def conv(self, x: int, weight: np.ndarray, bias: np.ndarray, act: str, *, stride: int = 1, pad: int = 0,
             dil: int = 1, out: Optional[int] = None, res: int = -1, gate_buf: int = -1, fbias_buf: int = -1,
             out_cs: int = 1, amax: Optional[Tuple[int, int, int]] = None, store_out: bool = True,
             cfg: int = -1, out_name: str = "", products: int = 3) -> int:
        """Dense conv as MFMA implicit GEMM.  weight [N,Cin,KH,KW] float (already BN-folded).
        ``products=1`` (split programs, opt-in): ONE f16 product per 32 k instead of the split's three, where the engine has such a
        kernel for the shape (csrc/k_hero.h; elsewhere the field is ignored and the conv runs at full precision)."""
        ti = self.tensors[x]
        n, cin, kh, kw = weight.shape
        assert cin == ti.real_c, (cin, ti.real_c)
        oh = (ti.H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        ow = (ti.W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        if out is None:
            out = self.tensor(oh, ow, _round_up(n, self.ve), name=out_name)
            self.tensors[out].real_c = n          # channels [n, C) are written as zeros by the epilogue
        to = self.tensors[out]
        assert (to.H, to.W) == (oh, ow), ((to.H, to.W), (oh, ow))
        woff, npad, cpad, acc_scale, use_split = self.pack_conv_weight(weight)
        b = np.zeros(npad, np.float64)
        b[:n] = bias
        boff = self.const_f32(b)
        av, ai, an = amax if amax is not None else (-1, -1, 0)
        self._op(OP_CONV, [x, out, woff, boff, res, gate_buf, fbias_buf, kh, kw, stride, pad, dil, cpad, npad, n,
                           ACT[act], out_cs, av, ai, an, 1 if store_out else 0, cfg,
                           struct.unpack("<i", struct.pack("<f", acc_scale))[0], (2 if products == 1 else 1) if use_split else 0],
                 [self._tb(x), self._tb(res), gate_buf, fbias_buf], [self._tb(out), av, ai])
        return out

    def dw(self, x: int, weight: np.ndarray, bias: np.ndarray, act: str, *, stride: int = 1, pad: int = 0,
           dil: int = 1, out_name: str = "") -> int:
        """Depthwise conv; weight [C,1,K,K] float (BN folded)."""
        ti = self.tensors[x]
        c, one, k, k2 = weight.shape
        assert one == 1 and k == k2 and c == ti.C
        oh = (ti.H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        ow = (ti.W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        out = self.tensor(oh, ow, c, name=out_name)
        woff = self.const_act(np.transpose(weight.astype(np.float64).reshape(c, k * k), (1, 0)))
        boff = self.const_f32(bias)
        self._op(OP_DW, [x, out, woff, boff, k, stride, pad, dil, ACT[act]], [self._tb(x)], [self._tb(out)])
        return out

    MBCONV_KERNELS = {(2, 1, 2), (1, 1, 2), (2, 2, 5), (1, 3, 5)}   # (stride, Cin/32, Cout/16 max) built in csrc/k_mbconv.h

    def mbconv_supported(self, cin: int, k: int, stride: int, dil: int, cout: int) -> bool:
        if self.esize != 4 or k != 3 or dil != 1 or cin % 8:
            return False
        return any(s == stride and ks == _round_up(cin, 32) // 32 and cout <= 16 * nt for s, ks, nt in self.MBCONV_KERNELS)

    @staticmethod
    def _split_rows(w: np.ndarray) -> Tuple[np.ndarray, float]:
        """[rows][K] (K % 32 == 0) float64 -> ([rows][K/32][hi 32 | lo 32] f16 of w * 2^s, 2^-s)."""
        rows, k = w.shape
        wmax = float(np.abs(w).max())
        s = 0 if wmax == 0.0 else int(np.floor(np.log2(16384.0 / wmax)))
        ws = (w * (2.0 ** s)).astype(np.float32)
        hi = ws.astype(np.float16)
        lo = (ws - hi.astype(np.float32)).astype(np.float16)
        blocks = np.stack([hi.reshape(rows, k // 32, 32), lo.reshape(rows, k // 32, 32)], axis=2)
        return np.ascontiguousarray(blocks), float(2.0 ** (-s))

    def _pow2_unscale(self, w: np.ndarray) -> float:
        """2^-s for weights the SPLIT flavour of the exact-f32 block kernels scales by 2^s and splits when it fetches them
        (same rule as _split_rows: the largest magnitude lands in [8192, 16384], so the low halves stay normal f16 numbers);
        1.0 in f32 / f16 programs, whose kernels use the weights as they are."""
        if not self.split:
            return 1.0
        wmax = float(np.abs(w).max())
        return 1.0 if wmax == 0.0 else float(2.0 ** (-int(np.floor(np.log2(16384.0 / wmax)))))

    def _f32_or_presplit(self, w: np.ndarray, unscale: float) -> int:
        """Weights of the exact-f32 block kernels (k_mbconv.h mbconv_wave_f32_kernel).  f32 programs: plain f32.  Split programs: the
        16 bytes of every four consecutive weights of a row hold their scaled split instead, [hi x 4 | lo x 4] f16 with
        x = w / unscale (a power of two: exact), hi = f16(x), lo = f16(x - hi) -- what the kernel used to compute in every wave."""
        w32 = np.ascontiguousarray(w, dtype=np.float32)
        if not self.split:
            return self.const_f32(w32)
        assert w32.ndim == 2 and w32.shape[1] % 4 == 0
        xs = w32 * np.float32(1.0 / unscale)
        hi = xs.astype(np.float16)
        lo = (xs - hi.astype(np.float32)).astype(np.float16)
        packed = np.concatenate([hi.reshape(w32.shape[0], -1, 4), lo.reshape(w32.shape[0], -1, 4)], axis=2)   # [rows][cols / 4][8]
        return self.const(np.ascontiguousarray(packed))

    def mbconv(self, x: int, w_exp: np.ndarray, b_exp: np.ndarray, w_dw: np.ndarray, b_dw: np.ndarray,
               w_pwl: np.ndarray, b_pwl: np.ndarray, act: str, *, stride: int, pad: int, dil: int = 1,
               res: int = -1, out_name: str = "") -> int:
        """Whole inverted-residual block (expand 1x1 -> depthwise 3x3 -> project 1x1 [+ x]) in one launch;
        weights BN-folded: w_exp [Mid,Cin,1,1], w_dw [Mid,1,3,3], w_pwl [Cout,Mid,1,1]."""
        ti = self.tensors[x]
        mid, cin = w_exp.shape[:2]
        cout, k = w_pwl.shape[0], w_dw.shape[2]
        assert cin == ti.real_c == ti.C and w_dw.shape == (mid, 1, k, k) and w_pwl.shape[1] == mid
        assert self.mbconv_supported(cin, k, stride, dil, cout)
        oh = (ti.H + 2 * pad - dil * (k - 1) - 1) // stride + 1
        ow = (ti.W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        out = self.tensor(oh, ow, cout, name=out_name)
        fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]
        mid16, coutp = _round_up(mid, 16), _round_up(cout, 16)
        if (stride, _round_up(cin, 16)) in ((2, 16), (1, 32)) and coutp <= 32:
            # high-resolution blocks: exact-f32 kernel (f32 fragments halve the register footprint; 16-channel steps)
            cp = _round_up(cin, 16)
            we = np.zeros((mid16, cp)); we[:mid, :cin] = w_exp.reshape(mid, cin)
            be = np.zeros(mid16); be[:mid] = b_exp
            wd = np.zeros((k * k, mid16)); wd[:, :mid] = w_dw.reshape(mid, k * k).T
            bd = np.zeros(mid16); bd[:mid] = b_dw
            wp = np.zeros((coutp, mid16)); wp[:cout, :mid] = w_pwl.reshape(cout, mid)
            bp = np.zeros(coutp); bp[:cout] = b_pwl
            se, sp = self._pow2_unscale(we), self._pow2_unscale(wp)
            self._op(OP_MBCONV, [x, out, res, self._f32_or_presplit(we, se), self.const_f32(be), self.const_f32(wd), self.const_f32(bd),
                                 self._f32_or_presplit(wp, sp), self.const_f32(bp), k, stride, pad, dil, ACT[act], mid16, cp, coutp, cout,
                                 mid16, fbits(se), fbits(sp), 1],
                     [self._tb(x), self._tb(res)], [self._tb(out)])
            return out
        midp, cp = _round_up(mid, 32), _round_up(cin, 32)
        we = np.zeros((midp, cp)); we[:mid, :cin] = w_exp.reshape(mid, cin)
        be = np.zeros(midp); be[:mid] = b_exp
        wd = np.zeros((k * k, midp)); wd[:, :mid] = w_dw.reshape(mid, k * k).T
        bd = np.zeros(midp); bd[:mid] = b_dw
        wp = np.zeros((coutp, midp)); wp[:cout, :mid] = w_pwl.reshape(cout, mid)
        bp = np.zeros(coutp); bp[:cout] = b_pwl
        we_s, se = self._split_rows(we)
        wp_s, sp = self._split_rows(wp)
        self._op(OP_MBCONV, [x, out, res, self.const(we_s), self.const_f32(be), self.const_f32(wd), self.const_f32(bd),
                             self.const(wp_s), self.const_f32(bp), k, stride, pad, dil, ACT[act], midp, cp // 32, coutp, cout,
                             mid16, fbits(se), fbits(sp), 0],
                 [self._tb(x), self._tb(res)], [self._tb(out)])
        return out

    def shuffle_unit_supported(self, c: int) -> bool:
        return self.split and c in (32, 64, 128)

    def shuffle_unit(self, x2: int, w1: np.ndarray, b1: np.ndarray, w_dw: np.ndarray, b_dw: np.ndarray, w2: np.ndarray,
                     b2: np.ndarray, act: str, out: int, out_cs: int, pass_src: int, pass_dst: int) -> int:
        """Stride-1 ShuffleNetV2 unit in one launch: act(1x1) -> depthwise 3x3 -> act(1x1) written to every out_cs-th
        channel of `out` (strided view), pass-through half copied to `pass_dst` (strided view); BN-folded weights."""
        ti = self.tensors[x2]
        c = w1.shape[0]
        assert self.shuffle_unit_supported(c) and w1.shape[:2] == (c, c) and w2.shape[:2] == (c, c) and ti.C == c
        assert w_dw.shape == (c, 1, 3, 3)
        to = self.tensors[out]
        assert (to.H, to.W) == (ti.H, ti.W)
        we_s, se = self._split_rows(w1.reshape(c, c).astype(np.float64))
        wp_s, sp = self._split_rows(w2.reshape(c, c).astype(np.float64))
        wd = w_dw.reshape(c, 9).T
        fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]
        self._op(OP_MBCONV, [x2, out, -1, self.const(we_s), self.const_f32(b1), self.const_f32(wd), self.const_f32(b_dw),
                             self.const(wp_s), self.const_f32(b2), 3, 1, 1, 1, ACT[act], c, c // 32, c, c, c, fbits(se), fbits(sp), 3,
                             ACT["none"], ACT[act], out_cs, pass_src, pass_dst],
                 [self._tb(x2), self._tb(pass_src)], [self._tb(out), self._tb(pass_dst)])
        return out

    def det_unit_supported(self, c: int, cin: int, stride: int) -> bool:
        """ShuffleV2Block as ONE workgroup-level launch (csrc/k_det.h det_unit_kernel): branch width c, block input cin."""
        return self.split and c in (32, 64, 128) and ((stride == 1 and cin == 2 * c) or (stride == 2 and cin in (16, 64, 128) and cin <= c))

    def det_unit(self, x: int, out: int, stride: int, w1: np.ndarray, b1: np.ndarray, w_dw: np.ndarray, b_dw: np.ndarray,
                 w2: np.ndarray, b2: np.ndarray, w_dw1: Optional[np.ndarray] = None, b_dw1: Optional[np.ndarray] = None,
                 w3: Optional[np.ndarray] = None, b3: Optional[np.ndarray] = None) -> int:
        """Whole ShuffleV2Block (yolov5-face models/common.py; oracle/detector_net.py::_shuffle_block), BN-folded weights, SiLU:
        stride 1: out[2i] = x[i], out[2i+1] = silu(pw2(dw(silu(pw1(x[c:])))))[i];
        stride 2: out[2i] = silu(pw3(dw1(x)))[i], out[2i+1] = branch 2 on all of x.  `out` is a 2c-channel view."""
        ti, to = self.tensors[x], self.tensors[out]
        c = w1.shape[0]
        cin = ti.C
        assert self.det_unit_supported(c, cin, stride) and to.C == 2 * c and ti.real_c == ti.C
        assert (to.H, to.W) == ((ti.H - 1) // stride + 1, (ti.W - 1) // stride + 1)
        cin2 = c if stride == 1 else cin
        k1 = _round_up(cin2, 32)
        assert w1.shape[:2] == (c, cin2) and w2.shape[:2] == (c, c) and w_dw.shape == (c, 1, 3, 3)
        fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]

        def rows(w, k):
            m = np.zeros((w.shape[0], k), np.float64)
            m[:, :w.shape[1]] = w.reshape(w.shape[0], -1)
            return self._split_rows(m)
        w1s, s1 = rows(w1, k1)
        w2s, s2 = rows(w2, c)
        f = [x, out, self.const(w1s), self.const_f32(b1), self.const_f32(w_dw.reshape(c, 9).T), self.const_f32(b_dw),
             self.const(w2s), self.const_f32(b2)]
        if stride == 2:
            assert w_dw1.shape == (cin, 1, 3, 3) and w3.shape[:2] == (c, cin)
            wd1 = np.zeros((9, k1)); wd1[:, :cin] = w_dw1.reshape(cin, 9).T
            bd1 = np.zeros(k1); bd1[:cin] = b_dw1
            w3s, s3 = rows(w3, k1)
            f += [self.const_f32(wd1), self.const_f32(bd1), self.const(w3s), self.const_f32(b3)]
        else:
            s3 = 1.0
            f += [-1, -1, -1, -1]
        f += [fbits(s1), fbits(s2), fbits(s3), c, k1, stride, cin]
        self._op(OP_DETUNIT, f, [self._tb(x)], [self._tb(out)])
        return out

    def det_stem(self, w1, b1, w2a, b2a, w2b, b2b, w3, b3, out_name: str = "") -> int:
        """yolov5-face StemBlock on the 3-channel program input in ONE launch (csrc/k_det.h det_stem_kernel), BN-folded weights:
        stem_1 [16,3,3,3] s2, stem_2a [8,16,1,1], stem_2b [16,8,3,3] s2, stem_3 [16,32,1,1] over cat(stem_2b, maxpool2x2(stem_1))."""
        assert self.split and w1.shape == (16, 3, 3, 3) and w2a.shape[:2] == (8, 16) and w2b.shape == (16, 8, 3, 3) and w3.shape[:2] == (16, 32)
        assert self.in_h % 4 == 0 and self.in_w % 4 == 0
        out = self.tensor(self.in_h // 4, self.in_w // 4, 16, name=out_name)
        fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]

        def rows(w, n, k):
            m = np.zeros((n, k), np.float64)
            m[:w.shape[0], :w.shape[1]] = w
            return self._split_rows(m)
        w1r = self._stem_k_order(w1)
        w1u, s1u = rows(w1r / 255.0, 16, 32)
        w1f, s1f = rows(w1r, 16, 32)
        wa, sa = rows(w2a.reshape(8, 16).astype(np.float64), 16, 32)
        wb, sb = rows(np.transpose(w2b.astype(np.float64), (0, 2, 3, 1)).reshape(16, 72), 16, 96)   # k = tap*8 + c
        wc, sc = rows(w3.reshape(16, 32).astype(np.float64), 16, 32)
        b2 = np.zeros(16); b2[:8] = b2a
        self._op(OP_DETSTEM, [out, self.const(w1u), self.const(w1f), self.const_f32(b1), self.const(wa), self.const_f32(b2), self.const(wb),
                              self.const_f32(b2b), self.const(wc), self.const_f32(b3), fbits(s1u), fbits(s1f), fbits(sa), fbits(sb), fbits(sc)],
                 [], [self._tb(out)])
        return out

    def hr_bottleneck_supported(self, x: int, mid: int, cout: int, has_ds: bool) -> bool:
        ti = self.tensors[x]
        return (self.split and mid == 64 and cout == 256 and ti.C == ti.real_c
                and ((ti.C == 64 and has_ds) or (ti.C == 256 and not has_ds)))

    def hr_bottleneck(self, x: int, w1, b1, w2, b2, w3, b3, wd=None, bd=None, out_name: str = "") -> int:
        """HRNet Bottleneck (timm hrnet.py; oracle/teacher_net.py::_bottleneck) in ONE launch (csrc/k_hrb.h): relu(conv3(relu(conv2_3x3(
        relu(conv1(x))))) + shortcut(x)), BN-folded weights; shortcut = x, or the 1x1 conv (wd, bd) of the first block of layer1."""
        ti = self.tensors[x]
        cin, mid, cout = ti.C, w1.shape[0], w3.shape[0]
        assert self.hr_bottleneck_supported(x, mid, cout, wd is not None)
        assert w1.shape[:2] == (mid, cin) and w2.shape == (mid, mid, 3, 3) and w3.shape[:2] == (cout, mid)
        out = self.tensor(ti.H, ti.W, cout, name=out_name)
        fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]
        w1s, s1 = self._split_rows(w1.reshape(mid, cin).astype(np.float64))
        w2off, npad, cpad, s2, _ = self.pack_conv_weight(w2, force_split=True)
        assert (npad, cpad) == (64, 64)
        w3s, s3 = self._split_rows(w3.reshape(cout, mid).astype(np.float64))
        f = [x, out, self.const(w1s), self.const_f32(b1), w2off, self.const_f32(b2), self.const(w3s), self.const_f32(b3)]
        sd = 1.0
        if wd is not None:
            assert wd.shape[:2] == (cout, cin)
            wds, sd = self._split_rows(wd.reshape(cout, cin).astype(np.float64))
            f += [self.const(wds), self.const_f32(bd)]
        else:
            f += [-1, -1]
        f += [fbits(s1), fbits(s2), fbits(s3), fbits(sd), cin]
        self._op(OP_HRB, f, [self._tb(x)], [self._tb(out)])
        return out

    @staticmethod
    def _stem_k_order(w: np.ndarray) -> np.ndarray:
        """[16,3,3,3] -> [16,32]: K order of the staged-image stem GEMMs (csrc/k_det.h det_stem_kernel, k_front.h): k groups 0..2 =
        (ky, j = kx*3 + ci < 8), group 3 = j = 8 of ky = 0..2, rest zero."""
        w9 = np.transpose(w.astype(np.float64), (0, 2, 3, 1)).reshape(w.shape[0], 3, 9)
        out = np.zeros((w.shape[0], 32), np.float64)
        for ky in range(3):
            out[:, 8 * ky:8 * ky + 8] = w9[:, ky, :8]
            out[:, 24 + ky] = w9[:, ky, 8]
        return out

    def front2_supported(self) -> bool:
        return self.split and self.in_h % 2 == 0 and self.in_w % 4 == 0

    def front2(self, w_stem, b_stem, act_stem: str, w_dw, b_dw, w_pw, b_pw, out_name: str = "") -> int:
        """conv_stem (3x3 s2, 3 -> 16, + act) + blocks.0.0 (depthwise 3x3 + relu -> 1x1 16 -> 16, + x) of the Student encoder on the
        program input in ONE launch (csrc/k_front2.h): the 16-channel stem map never reaches HBM.  BN-folded weights."""
        assert self.front2_supported()
        assert w_stem.shape == (16, 3, 3, 3) and w_dw.shape == (16, 1, 3, 3) and w_pw.shape[:2] == (16, 16)
        out = self.tensor(self.in_h // 2, self.in_w // 2, 16, name=out_name)
        fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]
        ws = self._stem_k_order(w_stem)
        wu, su = self._split_rows(ws / 255.0)
        wf, sf = self._split_rows(ws)
        self._op(OP_FRONT2, [out, self.const(wu), self.const(wf), self.const_f32(b_stem), fbits(su), fbits(sf), ACT[act_stem],
                             self.const_f32(w_dw.reshape(16, 9).T), self.const_f32(b_dw), self.const_f32(w_pw.reshape(16, 16)), self.const_f32(b_pw)],
                 [], [self._tb(out)])
        return out

    def det_c3_supported(self, cin: int, tail: str) -> bool:
        return self.split and (cin, tail) in ((192, "conv"), (128, "detect"))

    def det_c3(self, src_a: int, src_b: int, up_a: bool, w_cv1, b_cv1, w_cv2, b_cv2, w_m1, b_m1, w_m2, b_m2, w_cv3, b_cv3, *,
               out: int = -1, tail: str = "none", w_tail=None, b_tail=None, out2: int = -1, rows_buf: int = -1, row0: int = 0,
               det_stride: float = 0.0, anchors=None, nrows_total: int = 0) -> None:
        """C3 block (n = 1, shortcut = False; oracle/detector_net.py::_c3) on cat(src_a [nearest x2 upsampled if up_a], src_b) in ONE
        launch (csrc/k_det.h det_c3_kernel), BN-folded weights, SiLU everywhere.  tail "conv": + a 1x1 conv 64 -> 64 (silu) written
        to `out2`; tail "detect": + the Detect 1x1 conv (48 outputs, bias, no activation; raw values to `out2` if given) and its
        decode into rows [row0, row0 + 3 H W) of `rows_buf`."""
        ta = self.tensors[src_a]
        tb = self.tensors[src_b] if src_b >= 0 else None
        H, W = (ta.H * 2, ta.W * 2) if up_a else (ta.H, ta.W)
        cin = ta.C + (tb.C if tb else 0)
        assert self.det_c3_supported(cin, tail) and ta.C % 8 == 0 and (tb is None or (tb.H, tb.W) == (H, W))
        assert w_cv1.shape[:2] == (32, cin) and w_cv2.shape[:2] == (32, cin) and w_m1.shape[:2] == (32, 32) and w_m2.shape == (32, 32, 3, 3)
        assert w_cv3.shape[:2] == (64, 64)
        fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]
        wa, sa = self._split_rows(np.concatenate([w_cv1.reshape(32, cin), w_cv2.reshape(32, cin)], 0).astype(np.float64))
        wb, sb = self._split_rows(w_m1.reshape(32, 32).astype(np.float64))
        wc_off, npad, cpad, sc, _ = self.pack_conv_weight(w_m2, force_split=True)
        assert (npad, cpad) == (32, 32)
        wd, sd = self._split_rows(w_cv3.reshape(64, 64).astype(np.float64))
        f = [src_a, src_b, out, out2, rows_buf, self.const(wa), self.const_f32(np.concatenate([b_cv1, b_cv2])), self.const(wb), self.const_f32(b_m1),
             wc_off, self.const_f32(b_m2), self.const(wd), self.const_f32(b_cv3)]
        se = 1.0
        if tail == "conv":
            assert w_tail.shape[:2] == (64, 64) and out2 >= 0
            we, se = self._split_rows(w_tail.reshape(64, 64).astype(np.float64))
            f += [self.const(we), self.const_f32(b_tail), -1]
        else:
            assert w_tail.shape[:2] == (48, 64) and rows_buf >= 0
            we, se = self._split_rows(w_tail.reshape(48, 64).astype(np.float64))
            f += [self.const(we), self.const_f32(b_tail), self.const_f32(np.asarray(anchors, np.float64).reshape(-1))]
        f += [fbits(sa), fbits(sb), fbits(sc), fbits(sd), fbits(se), fbits(float(det_stride)), cin, {"conv": 1, "detect": 2}[tail],
              1 if up_a else 0, row0, nrows_total]
        self._op(OP_DETC3, f, [self._tb(src_a), self._tb(src_b)], [self._tb(out), self._tb(out2), rows_buf])

    def dsconv_supported(self, cin: int, k: int, stride: int, dil: int, cout: int) -> bool:
        return self.esize == 4 and cin == 16 and k == 3 and stride == 1 and dil == 1 and cout <= 32 and cout % 4 == 0

    def dsconv(self, x: int, w_dw: np.ndarray, b_dw: np.ndarray, w_pw: np.ndarray, b_pw: np.ndarray, act: str, *,
               res: int = -1, out_name: str = "") -> int:
        """Depthwise-separable block (depthwise 3x3 + act -> pointwise [+ x]) in one launch; BN-folded weights."""
        ti = self.tensors[x]
        cout, cin = w_pw.shape[:2]
        assert cin == ti.real_c == ti.C and w_dw.shape == (cin, 1, 3, 3) and self.dsconv_supported(cin, 3, 1, 1, cout)
        out = self.tensor(ti.H, ti.W, cout, name=out_name)
        coutp = _round_up(cout, 16)
        wd = w_dw.reshape(cin, 9).T
        wp = np.zeros((coutp, cin)); wp[:cout] = w_pw.reshape(cout, cin)
        bp = np.zeros(coutp); bp[:cout] = b_pw
        fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]
        zero = self.const_f32(np.zeros(16))
        self._op(OP_MBCONV, [x, out, res, zero, zero, self.const_f32(wd), self.const_f32(b_dw), self.const_f32(wp),
                             self.const_f32(bp), 3, 1, 1, 1, ACT[act], 16, 16, coutp, cout, 16, fbits(1.0), fbits(1.0), 2],
                 [self._tb(x), self._tb(res)], [self._tb(out)])
        return out

    def expdw_supported(self, H: int, W: int, k: int, stride: int, pad: int, dil: int, cin: int = 0) -> bool:
        if (H, W) == (64, 64) and stride == 2:   # expdw_image_s2_kernel: 64x64 -> 32x32, 5x5, <= 32 input channels
            return self.split and 0 < cin <= 32 and cin % 8 == 0 and (k, dil, pad) == (5, 1, 2)
        if (H, W) == (32, 32):      # whole-image kernel (expdw_image_kernel): <= 64 input channels
            return (self.split and stride == 1 and 0 < cin <= 64 and cin % 8 == 0 and (k, dil) in ((3, 1), (5, 1))
                    and pad == dil * (k - 1) // 2)
        ohw = H * W
        return (self.split and stride == 1 and W <= 16 and 256 % ohw == 0 and 256 // ohw <= 4
                and (k, dil) in ((3, 1), (5, 1), (5, 2)) and pad == dil * (k - 1) // 2)

    def expdw(self, x: int, w_exp: np.ndarray, b_exp: np.ndarray, w_dw: np.ndarray, b_dw: np.ndarray, act: str, *,
              pad: int, dil: int = 1, stride: int = 1, want_gap: bool = False, out_name: str = "") -> Tuple[int, int]:
        """Pointwise expand + depthwise kxk (+BN, act each) in one launch; the expanded tensor stays in LDS.
        Returns (depthwise output tensor, buffer with its per-face channel means or -1)."""
        ti = self.tensors[x]
        mid, cin = w_exp.shape[:2]
        k = w_dw.shape[2]
        assert cin == ti.real_c and w_dw.shape == (mid, 1, k, k) and mid % self.ve == 0
        assert self.expdw_supported(ti.H, ti.W, k, stride, pad, dil, cin)
        out = self.tensor(ti.H // stride, ti.W // stride, mid, name=out_name)
        woff, npad, cpad, acc_scale, use_split = self.pack_conv_weight(w_exp, force_split=True)
        be = np.zeros(npad, np.float64)
        be[:mid] = b_exp
        gap = self.buffer(mid, ELEM_F32, "gap") if want_gap else -1
        wd = np.transpose(w_dw.astype(np.float64).reshape(mid, k * k), (1, 0))
        self._op(OP_EXPDW, [x, out, gap, woff, self.const_f32(be), self.const_f32(wd), self.const_f32(b_dw), k, pad, dil,
                            ACT[act], cpad, npad, mid, struct.unpack("<i", struct.pack("<f", acc_scale))[0], stride],
                 [self._tb(x)], [self._tb(out), gap])
        return out, gap

    # (Cin padded / 32, Cout / 16, k, dilation, has SE) with an mbx_kernel instantiation (csrc/k_mbx.h, engine.cpp PF_OP_MBX)
    MBX_KERNELS = {(3, 5, 3, 1, False), (3, 7, 3, 1, True), (4, 7, 3, 1, True), (4, 10, 5, 1, True), (5, 10, 5, 2, True)}

    def mbx_supported(self, x: int, k: int, stride: int, pad: int, dil: int, cout: int, se: bool) -> bool:
        """A whole inverted-residual block at 16 x 16 in the input-stationary kernel (csrc/k_mbx.h): f32s programs, stride 1."""
        ti = self.tensors[x]
        return (self.split and (ti.H, ti.W) == (16, 16) and stride == 1 and pad == dil * (k - 1) // 2 and ti.C == ti.real_c
                and ti.C % 4 == 0 and cout % 16 == 0 and (_round_up(ti.C, 32) // 32, cout // 16, k, dil, bool(se)) in self.MBX_KERNELS)

    def _mbx_pack(self, w_exp, b_exp, w_dw, b_dw, w_pwl, b_pwl):
        """(w1 off, ctile off, w2 off, b2 off, KS, T, scale1, scale2): expand rows in tiles of 32 expanded channels (zero rows up
        to a whole tile), per-tile constants [T][k*k + 2][32] = depthwise taps | expand bias | depthwise bias, projection rows
        [Cout][T][hi 32 | lo 32]."""
        mid, cin = w_exp.shape[:2]
        cout, k = w_pwl.shape[0], w_dw.shape[2]
        T, cp = _round_up(mid, 32) // 32, _round_up(cin, 32)
        we = np.zeros((T * 32, cp)); we[:mid, :cin] = w_exp.reshape(mid, cin)
        wp = np.zeros((cout, T * 32)); wp[:, :mid] = w_pwl.reshape(cout, mid)
        ct = np.zeros((T, k * k + 2, 32))
        taps = np.zeros((k * k, T * 32)); taps[:, :mid] = w_dw.reshape(mid, k * k).T
        be = np.zeros(T * 32); be[:mid] = b_exp
        bd = np.zeros(T * 32); bd[:mid] = b_dw
        ct[:, :k * k, :] = taps.reshape(k * k, T, 32).transpose(1, 0, 2)
        ct[:, k * k, :] = be.reshape(T, 32)
        ct[:, k * k + 1, :] = bd.reshape(T, 32)
        we_s, s1 = self._split_rows(we)
        wp_s, s2 = self._split_rows(wp)
        return self.const(we_s), self.const_f32(ct), self.const(wp_s), self.const_f32(b_pwl), cp // 32, T, s1, s2

    MBX_RECOMPUTE_16 = {(3, 7, 3, 1), (4, 7, 3, 1)}     # (KS, Cout / 16, k, dil) whose gate-and-project pass fits 16 waves x 128 registers

    def mbx(self, x: int, w_exp, b_exp, w_dw, b_dw, w_pwl, b_pwl, act: str, *, pad: int, dil: int = 1, res: int = -1,
            se_fcs=None, se_mode: Optional[str] = None, waves: int = 16, out_name: str = "", dw_name: str = "") -> int:
        """Inverted-residual block (expand 1x1 -> depthwise kxk -> [SE] -> project 1x1 [+ res]) on a 16 x 16 map, weights BN-folded.
        Without SE: one launch.  With SE (``se_fcs`` = (w_reduce [R,Mid], b_reduce, w_expand [Mid,R], b_expand)) a squeeze pass
        (expand + depthwise -> per-face channel means), the two FCs, then by ``se_mode``
          "store":     the squeeze pass also stores the activated depthwise map and the layer-wise gated projection reads it
                       back (the default: measured faster for every SE block of the Student, profiles/r05_run11_mbx_ab_v4b.txt --
                       the depthwise is VALU-bound, so a second pass over it costs more than the map's round trip through HBM);
          "recompute": a second pass that recomputes expand + depthwise, applies the gate and projects -- the expanded tensor
                       never exists in HBM (0.134 against 0.097 ms per 256 faces for the cheapest case, 80 -> 480 -> 112 with 3 x 3).
        ``waves``: 16 or 8 waves per workgroup for the launches that have both flavours (A/B aid)."""
        ti = self.tensors[x]
        mid, cin = w_exp.shape[:2]
        cout, k = w_pwl.shape[0], w_dw.shape[2]
        assert cin == ti.real_c == ti.C and w_dw.shape == (mid, 1, k, k) and w_pwl.shape[1] == mid
        assert self.mbx_supported(x, k, 1, pad, dil, cout, se_fcs is not None) and waves in (8, 16)
        w1, ct, w2, b2, ks, T, s1, s2 = self._mbx_pack(w_exp, b_exp, w_dw, b_dw, w_pwl, b_pwl)
        fbits = lambda v: struct.unpack("<i", struct.pack("<f", v))[0]
        common = [w1, ct, w2, b2, k, pad, dil, ACT[act], ks, T, cout, mid, fbits(s1), fbits(s2)]
        if se_fcs is None:
            out = self.tensor(ti.H, ti.W, cout, name=out_name)
            self._op(OP_MBX, [x, out, res, -1, -1] + common + [0, waves], [self._tb(x), self._tb(res)], [self._tb(out)])
            return out
        if se_mode is None:
            se_mode = "store"
        assert se_mode in ("recompute", "store")
        w_rd, b_rd, w_ex, b_ex = se_fcs
        gap = self.buffer(mid, ELEM_F32, "gap")
        if se_mode == "store":
            assert mid % 32 == 0, "the stored map is written in whole 32-channel tiles"
            dwt = self.tensor(ti.H, ti.W, mid, name=dw_name)
            self._op(OP_MBX, [x, dwt, -1, gap, -1] + common + [3, 16], [self._tb(x)], [self._tb(dwt), gap])
            gate = self.fc_pair(gap, w_rd, b_rd, "relu", w_ex, b_ex, "hsigmoid")
            return self.conv(dwt, w_pwl, b_pwl, "none", res=res, gate_buf=gate, out_name=out_name)
        assert mid % 32 == 0, "the recompute pass fetches the face's gates in whole 32-channel tiles (k_mbx.h dma_ct)"
        out = self.tensor(ti.H, ti.W, cout, name=out_name)
        self._op(OP_MBX, [x, -1, -1, gap, -1] + common + [1, 16], [self._tb(x)], [gap])
        gate = self.fc_pair(gap, w_rd, b_rd, "relu", w_ex, b_ex, "hsigmoid")
        nw2 = waves if (ks, cout // 16, k, dil) in self.MBX_RECOMPUTE_16 else 8
        self._op(OP_MBX, [x, out, res, -1, gate] + common + [2, nw2], [self._tb(x), self._tb(res), gate], [self._tb(out)])
        return out

    CHAIN_SHAPES = ((72, 16), (144, 8))       # (channels, map side) with a basic_chain_kernel instantiation
    CHAIN_MAX_CONVS = 8

    def basic_chain_supported(self, x: int, n_blocks: int) -> bool:
        ti = self.tensors[x]
        return (self.split and ti.H == ti.W and (ti.real_c, ti.H) in self.CHAIN_SHAPES and ti.C == ti.real_c
                and 1 <= n_blocks <= self.CHAIN_MAX_CONVS // 2)

    def basic_chain(self, x: int, blocks: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]], out_name: str = "") -> int:
        """n BasicBlocks -- relu(conv2(relu(conv1(x))) + x), BN folded, both convs 3x3 / stride 1 / pad 1, C -> C -- in one
        launch with the face's map resident in LDS (k_chain.h).  blocks: (w1, b1, w2, b2) per block."""
        ti = self.tensors[x]
        assert self.basic_chain_supported(x, len(blocks))
        c = ti.real_c
        out = self.tensor(ti.H, ti.W, c, name=out_name)
        fields = [x, out, 2 * len(blocks), c]
        for blk in blocks:
            for wgt, bias in ((blk[0], blk[1]), (blk[2], blk[3])):
                assert wgt.shape == (c, c, 3, 3)
                woff, npad, cpad, acc_scale, _ = self.pack_conv_weight(wgt, force_split=True)
                assert npad == _round_up(c, 16) and cpad == _round_up(c, 32)
                b = np.zeros(npad, np.float64)
                b[:c] = bias
                fields += [woff, self.const_f32(b), struct.unpack("<i", struct.pack("<f", acc_scale))[0]]
        self._op(OP_CHAIN, fields, [self._tb(x)], [self._tb(out)])
        return out

    BLOCK_SHAPES = ((18, 64), (36, 32), (18, 16))   # (channels, map side) with a basic_block_kernel instantiation

    def basic_block_supported(self, x: int) -> bool:
        ti = self.tensors[x]
        return self.split and ti.H == ti.W and (ti.real_c, ti.H) in self.BLOCK_SHAPES and ti.C == _round_up(ti.real_c, 4)

    def pack_flatk_weight(self, weight: np.ndarray) -> Tuple[int, int, float]:
        """[N,C,3,3] -> (const offset, Npad, acc_scale): split weights with the (tap, 8-channel group) axis flattened,
        [Npad][NCH][hi 32 x f16 | lo 32 x f16] of w * 2^s, k = tap * 8 * ceil(C/8) + c, NCH = ceil(9 * ceil(C/8) / 4)."""
        n, c, kh, kw = weight.shape
        assert (kh, kw) == (3, 3)
        cg = (c + 7) // 8
        nch = (9 * cg + 3) // 4
        npad = _round_up(n, 16)
        w = np.zeros((npad, 9, cg * 8), np.float64)
        w[:n, :, :c] = np.transpose(weight.astype(np.float64), (0, 2, 3, 1)).reshape(n, 9, c)
        flat = np.zeros((npad, nch * 32), np.float64)
        flat[:, :9 * cg * 8] = w.reshape(npad, 9 * cg * 8)
        wmax = float(np.abs(flat).max())
        s = 0 if wmax == 0.0 else int(np.floor(np.log2(16384.0 / wmax)))
        ws = (flat * (2.0 ** s)).astype(np.float32)
        hi = ws.astype(np.float16)
        lo = (ws - hi.astype(np.float32)).astype(np.float16)
        blocks = np.stack([hi.reshape(npad, nch, 32), lo.reshape(npad, nch, 32)], axis=2)
        return self.const(np.ascontiguousarray(blocks)), npad, float(2.0 ** (-s))

    def basic_block(self, x: int, w1: np.ndarray, b1: np.ndarray, w2: np.ndarray, b2: np.ndarray, out_name: str = "") -> int:
        """relu(conv2(relu(conv1(x))) + x), BN folded, 3x3 / stride 1 / pad 1, C -> C, one launch (k_chain.h basic_block_kernel)."""
        ti = self.tensors[x]
        assert self.basic_block_supported(x)
        c = ti.real_c
        out = self.tensor(ti.H, ti.W, ti.C, name=out_name)
        self.tensors[out].real_c = c
        fields = [x, out, c]
        for wgt, bias in ((w1, b1), (w2, b2)):
            assert wgt.shape == (c, c, 3, 3)
            woff, npad, acc_scale = self.pack_flatk_weight(wgt)
            b = np.zeros(npad, np.float64)
            b[:c] = bias
            fields += [woff, self.const_f32(b), struct.unpack("<i", struct.pack("<f", acc_scale))[0]]
        self._op(OP_BLOCK, fields, [self._tb(x)], [self._tb(out)])
        return out

    def upcat(self, lo: int, skip: int, out_name: str = "") -> int:
        tl, ts = self.tensors[lo], self.tensors[skip]
        assert (ts.H, ts.W) == (2 * tl.H, 2 * tl.W)
        out = self.tensor(ts.H, ts.W, tl.C + ts.C, name=out_name)
        self._op(OP_UPCAT, [lo, skip, out], [self._tb(lo), self._tb(skip)], [self._tb(out)])
        return out

    def sepconv_up_can_sum(self, lo: int, skip: int, n_out: int) -> bool:
        """Will ``sepconv_up`` run as the pipelined kernel's 256-output instance (csrc/k_sepup.h, engine.cpp pipe_ok), whose consumers can
        leave per-tile channel sums of the output behind (``gap_parts=True``)?"""
        tl, ts = self.tensors[lo], self.tensors[skip]
        return (self.split and n_out == 256 and ts.W in (16, 32, 64) and (ts.H * ts.W) % 128 == 0 and (ts.H * ts.W) // 128 >= 2
                and tl.C % 32 == 0 and ts.C % 8 == 0 and ts.C <= 64 and _round_up(tl.C + ts.C, 32) <= 640)

    def sepconv_up(self, lo: int, skip: int, dw_weight: np.ndarray, dw_bias: np.ndarray, pw_weight: np.ndarray,
                   pw_bias: np.ndarray, act: str, out_name: str = "", gap_parts: bool = False):
        """Fused DecoderBlock front end: cat(bilinear_x2(lo), skip) -> depthwise 3x3 (+bias, BN folded) ->
        1x1 conv (+bias, BN folded) -> act, in one split-precision GEMM launch (f32s programs only)."""
        assert self.split
        tl, ts = self.tensors[lo], self.tensors[skip]
        c = tl.C + ts.C
        n, cin, kh, kw = pw_weight.shape
        assert (ts.H, ts.W) == (2 * tl.H, 2 * tl.W) and cin == c and kh == kw == 1 and tl.C % 32 == 0
        assert dw_weight.shape == (c, 1, 3, 3) and self.conv_uses_split(c)
        out = self.tensor(ts.H, ts.W, _round_up(n, self.ve), name=out_name)
        woff, npad, cpad, acc_scale, use_split = self.pack_conv_weight(pw_weight)
        assert use_split
        # Only the affine BatchNorm sits between the depthwise and the pointwise conv (SeparableConv2d, model.py:21-30), so
        # the depthwise bias is a constant vector in front of a linear map: it moves into the pointwise bias here
        # (float64) and the kernels' producers start from zero -- no per-K-step bias traffic.
        b = np.zeros(npad, np.float64)
        b[:n] = pw_bias.astype(np.float64) + pw_weight.astype(np.float64).reshape(n, c) @ dw_bias.astype(np.float64)
        dw_bias = np.zeros(c, np.float64)
        wdw = dw_weight.astype(np.float64).reshape(c, 3, 3)
        c1 = tl.C
        assert ts.H >= 6 and ts.W >= 6
        # upsample (bilinear x2, align_corners=False) followed by the zero-padded depthwise 3x3 == one 3x3
        # filter on the low-res map with position-class dependent weights  E = A^T . W . B
        ay = [_upsample_tap_matrix(y, ts.H, tl.H) for y in (0, ts.H - 1, 2, 3)]   # classes first/last/even/odd
        ax = [_upsample_tap_matrix(x, ts.W, tl.W) for x in (0, ts.W - 1, 2, 3)]
        E = np.zeros((4, 4, 3, 3, c1), np.float64)
        for cy in range(4):
            for cx in range(4):
                # E[j,i,c] = sum_ky sum_kx A[ky,j] * B[kx,i] * W[c,ky,kx]
                E[cy, cx] = np.einsum("kj,li,ckl->jic", ay[cy], ax[cx], wdw[:c1])
        dwe = self.const_f32(E.reshape(16 * 9, c1))
        dws = self.const_f32(np.transpose(wdw[c1:].reshape(c - c1, 9), (1, 0)))
        # scratch of the pipelined kernel (csrc/k_sepup.h): the skip channels' depthwise output, pre-split, one 16 KB pixel-operand
        # stage per (128-pixel tile, 32-channel chunk); lives for this op only
        n_skip_chunks = cpad // 32 - c1 // 32
        skipx = self.buffer((ts.H * ts.W + 127) // 128 * n_skip_chunks * 4096, ELEM_F32, "sepup.skipx")
        dwl = self.const_f32(np.transpose(wdw[:c1].reshape(c1, 9), (1, 0)))       # [9][C1]: plain filters (upsample, then filter)
        # round 6: the pipelined kernel interpolates HORIZONTALLY only; the vertical half of the upsample and the zero rows above /
        # below the image are folded into the filter, one 3 x 3 set per row class (first / last / even / odd):
        # V[cy][j][kx][c] = sum_ky A_cy[ky][j] * W[c][ky][kx]  (csrc/k_sepup.h, VCOL)
        V = np.stack([np.einsum("kj,ckl->jlc", ay[cy], wdw[:c1]) for cy in range(4)])
        dwv = self.const_f32(V.reshape(4 * 9, c1))
        parts = -1
        if gap_parts:       # per-tile (128 pixels) channel sums of the activated output: the squeeze of the SCSE block behind it, for fc_pair(nparts=...)
            assert self.sepconv_up_can_sum(lo, skip, n) and n == npad
            parts = self.buffer((ts.H * ts.W // 128) * n, ELEM_F32, "sepup.gap_parts")
        self._op(OP_SEPUP, [lo, skip, out, dwe, self.const_f32(dw_bias), woff, self.const_f32(b), cpad, npad, n, ACT[act],
                            struct.unpack("<i", struct.pack("<f", acc_scale))[0], dws, skipx, dwl, dwv, parts + 1],
                 [self._tb(lo), self._tb(skip)], [self._tb(out), skipx, parts])
        return (out, parts) if gap_parts else out

    def add_up(self, a: int, b: int, shift: int, act: str, out_name: str = "") -> int:
        """out = act(a + nearest_upsample(b, 2**shift)) (HRNet fuse layers)."""
        ta, tb = self.tensors[a], self.tensors[b]
        assert (tb.H << shift, tb.W << shift, tb.C) == (ta.H, ta.W, ta.C)
        out = self.tensor(ta.H, ta.W, ta.C, name=out_name)
        self.tensors[out].real_c = ta.real_c
        self._op(OP_ADDUP, [a, b, out, shift, ACT[act]], [self._tb(a), self._tb(b)], [self._tb(out)])
        return out

    def fuse_up_supported(self, y: int, srcs) -> bool:
        ty = self.tensors[y]
        if self.esize != 4 or not 1 <= len(srcs) <= 3 or ty.C % 4:
            return False
        need = 0
        for t, shift in srcs:
            ts = self.tensors[t]
            if not 1 <= shift <= 3 or (ts.H << shift, ts.W << shift) != (ty.H, ty.W) or ts.C % 4 or ts.C != ts.real_c:
                return False
            r = max(1, 16 >> shift)
            need += ts.C * ty.C + r * r * (ts.C + ty.C)
        return need <= 24576

    def fuse_up(self, y: int, terms, act: str, out_name: str = "") -> int:
        """out = act(y + sum_s nearest_upsample(conv1x1_s(src_s) + b_s, 2**shift_s)) in ONE launch (csrc/k_layers.h fuse_up_kernel): the
        fuse sum of an HRNet module towards one of its higher-resolution branches (timm hrnet.py fuse_layers[i][j > i]).
        terms = [(src tensor, weight [C, srcC, 1, 1] BN-folded, bias [C], shift)], added in the order given."""
        ty = self.tensors[y]
        assert self.fuse_up_supported(y, [(t, sh) for t, _, _, sh in terms])
        out = self.tensor(ty.H, ty.W, ty.C, name=out_name)
        self.tensors[out].real_c = ty.real_c
        f = [y, out, ACT[act], len(terms)]
        reads = [self._tb(y)]
        for t, w, b, shift in terms:
            ts = self.tensors[t]
            c, k = w.shape[0], w.shape[1]
            assert c == ty.real_c and k == ts.real_c == ts.C
            wt = np.zeros((k, ty.C), np.float64)
            wt[:, :c] = w.reshape(c, k).T
            bb = np.zeros(ty.C, np.float64)
            bb[:c] = b
            f += [t, self.const_f32(wt), self.const_f32(bb), shift]
            reads.append(self._tb(t))
        f += [-1, 0, 0, 0] * (3 - len(terms)) + [ty.real_c]
        self._op(OP_FUSEUP, f, reads, [self._tb(out)])
        return out

    def gap(self, x: int) -> int:
        out = self.buffer(self.tensors[x].C, ELEM_F32, "gap")
        self._op(OP_GAP, [x, out], [self._tb(x)], [out])
        return out

    def fc(self, xbuf: int, weight: np.ndarray, bias: Optional[np.ndarray], act: str,
           scale2: Optional[np.ndarray] = None, shift2: Optional[np.ndarray] = None, act2: str = "none") -> int:
        """y = act(W x + b) on pooled f32 vectors; weight [N,K]."""
        n, k = weight.shape
        assert self.bufs[xbuf].elems == k
        out = self.buffer(n, ELEM_F32, "fc")
        woff = self.const_f32(np.transpose(weight.astype(np.float64), (1, 0)))
        boff = self.const_f32(bias) if bias is not None else -1
        s2 = self.const_f32(scale2) if scale2 is not None else -1
        t2 = self.const_f32(shift2) if shift2 is not None else -1
        self._op(OP_FC, [xbuf, out, woff, boff, k, n, ACT[act], s2, t2, ACT[act2]], [xbuf], [out])
        return out

    def fc_pair_fuses(self, k: int, r: int, n: int) -> bool:
        # one launch streams BOTH matrices through every workgroup (4 faces each): a win while they are small -- 5.4 against 9.0 us for
        # 72 -> 24 -> 72, 11.1 against 12.9 us for 480 -> 120 -> 480 -- and a loss once a compute unit's 64 B / clock from the L2 is the
        # bound (960 -> 240 -> 960, 1.8 MB per workgroup: 23.7 against 20.6 us; profiles/r06_run10_ub_fc2.txt).
        # f32s programs only: the exact-f32 programs keep the two-launch form and with it their bit pattern (the golden of
        # tests/test_tracking_parity.py was produced by the f32 engine as the reference FaceAna's landmark session)
        return bool(self.split and getattr(self, "fuse_fc_pairs", True) and r % 4 == 0 and n % 4 == 0 and max(k, r, n) <= 960 and k * r + r * n <= 131072)

    def fc_pair(self, xbuf: int, w1: np.ndarray, b1: Optional[np.ndarray], act1: str, w2: np.ndarray, b2: Optional[np.ndarray], act2: str,
                scale2: Optional[np.ndarray] = None, shift2: Optional[np.ndarray] = None, act1b: str = "none", nparts: int = 1,
                xscale: float = 1.0) -> int:
        """y = act2(W2 h + b2), h = act1b(scale2 * act1(W1 x + b1) + shift2): two dependent FCs on pooled vectors -- an SE gate, the
        cSE gate, the ASPP's pooled branch.  ONE launch (csrc/k_layers.h fc2_kernel) when ``self.fuse_fc_pairs`` and the shapes allow
        it (R, N multiples of 4, everything <= 960), two ``fc`` ops otherwise."""
        r, k = w1.shape
        n, r2 = w2.shape
        assert r2 == r and self.bufs[xbuf].elems == k * nparts
        if not self.fc_pair_fuses(k, r, n):
            assert nparts == 1, "partial-sum inputs need the fused launch"
            hid = self.fc(xbuf, w1, b1, act1, scale2=scale2, shift2=shift2, act2=act1b)
            return self.fc(hid, w2, b2, act2)
        out = self.buffer(n, ELEM_F32, "fc2")
        cf = lambda v: self.const_f32(v) if v is not None else -1
        self._op(OP_FC2, [xbuf, out, self.const_f32(np.transpose(w1.astype(np.float64), (1, 0))), cf(b1), k, r, ACT[act1],
                          cf(scale2), cf(shift2), ACT[act1b], self.const_f32(np.transpose(w2.astype(np.float64), (1, 0))), cf(b2), n, ACT[act2],
                          nparts, struct.unpack("<i", struct.pack("<f", float(xscale)))[0]],
                 [xbuf], [out])
        return out

    def scse(self, x: int, cse_buf: int, sse_w: np.ndarray, sse_b: float, out_name: str = "") -> int:
        ti = self.tensors[x]
        out = self.tensor(ti.H, ti.W, ti.C, name=out_name)
        woff = self.const_f32(sse_w.reshape(-1))
        bbits = struct.unpack("<i", struct.pack("<f", float(sse_b)))[0]
        self._op(OP_SCSE, [x, out, cse_buf, woff, bbits], [self._tb(x), cse_buf], [self._tb(out)])
        return out

    def hmdec(self, val_buf: int, idx_buf: int, feat: int, off_w: np.ndarray, off_b: np.ndarray, points: int,
              nslots: int) -> Tuple[int, int]:
        loc = self.buffer(points * 2, ELEM_F32, "loc_fix", pinned=True)
        score = self.buffer(points, ELEM_F32, "score", pinned=True)
        woff = self.const_f32(off_w)
        boff = self.const_f32(off_b)
        self._op(OP_HMDEC, [val_buf, idx_buf, feat, woff, boff, points, nslots, loc, score],
                 [val_buf, idx_buf, self._tb(feat)], [loc, score])
        return loc, score

    def maxpool(self, x: int, out: Optional[int] = None, out_name: str = "") -> int:
        ti = self.tensors[x]
        oh, ow = (ti.H + 1) // 2, (ti.W + 1) // 2
        if out is None:
            out = self.tensor(oh, ow, ti.C, name=out_name)
        self._op(OP_MAXPOOL, [x, out], [self._tb(x)], [self._tb(out)])
        return out

    def copy(self, x: int, out: int, out_cs: int = 1, up: int = 1):
        self._op(OP_COPY, [x, out, out_cs, up], [self._tb(x)], [self._tb(out)])

    def detdec(self, x: int, rows_buf: int, row0: int, stride: float, anchors: np.ndarray, nrows_total: int):
        aoff = self.const_f32(np.asarray(anchors, np.float64).reshape(-1))
        sbits = struct.unpack("<i", struct.pack("<f", float(stride)))[0]
        self._op(OP_DETDEC, [x, rows_buf, row0, sbits, aoff, nrows_total], [self._tb(x)], [rows_buf])

    # ---- assembly ---------------------------------------------------------------------------
    def _item_units(self, b: _Buf) -> int:
        es = self.esize if b.etype == ELEM_ACT else (1 if b.etype == ELEM_U8 else 4)
        return max(1, (b.elems * es + 255) // 256)

    def _plan(self) -> int:
        for oi, (_, _, reads, writes) in enumerate(self.ops):
            for b in reads + writes:
                bb = self.bufs[b]
                bb.first = oi if bb.first is None else bb.first
                bb.last = oi
        n_ops = len(self.ops)
        for b in self.bufs:
            if b.first is None:
                b.first, b.last = 0, n_ops
            if b.pinned or self.keep_all:
                b.first, b.last = 0, n_ops
        # greedy first-fit over buffers sorted by first use; intervals [first, last]
        placed: List[_Buf] = []
        total = 0
        for b in sorted(self.bufs, key=lambda q: (q.first, -self._item_units(q))):
            size = self._item_units(b)
            busy = sorted((p.offset_units, p.offset_units + self._item_units(p)) for p in placed
                          if not (p.last < b.first or p.first > b.last))
            off = 0
            for lo, hi in busy:
                if off + size <= lo:
                    break
                off = max(off, hi)
            b.offset_units = off
            placed.append(b)
            total = max(total, off + size)
        return total

    def finish(self, out_bufs: Sequence[int]) -> bytes:
        for ob in out_bufs:
            self.bufs[ob].pinned = True
        arena_units = self._plan()
        outs = list(out_bufs) + [-1] * (3 - len(out_bufs))
        blob = bytearray()
        blob += struct.pack("<16i", MAGIC, VERSION, self.dtype, len(self.bufs), len(self.tensors), len(self.ops),
                            len(self.consts), arena_units, self.in_h, self.in_w, outs[0], outs[1], outs[2], 0, 0, 0)
        for b in self.bufs:
            blob += struct.pack("<4i", b.etype, b.elems, b.offset_units, 0)
        for t in self.tensors:
            blob += struct.pack("<8i", t.buf, t.coff, t.ld, t.H, t.W, t.C, 0, 0)
        for code, f, _, _ in self.ops:
            blob += struct.pack(f"<{OP_FIELDS + 1}i", code, *f)
        while len(blob) % 256:
            blob.append(0)
        blob += self.consts
        return bytes(blob)
