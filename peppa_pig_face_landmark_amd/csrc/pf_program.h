// Packed network program: the binary interface between the Python graph builder
// (peppa_pig_face_landmark_amd/graph/ir.py -- keep the two in sync) and the HIP executor.
//
// A program is a straight-line list of fused layer ops over NHWC activation tensors that live in
// one device arena.  Everything is little-endian int32 (floats are bit-cast), so the Python side
// needs nothing but struct.pack.
//
//   blob := Header | BufRec[n_bufs] | TensorRec[n_tensors] | OpRec[n_ops] | pad to 256 | const bytes
//
// Buffers scale linearly with the batch: buffer i lives at arena + offset_units*256*max_batch and
// item b of it at + b * item_bytes.  Tensors are (buffer, channel offset, pixel stride) views, which
// is how torch.cat (DecoderBlock, ASPP, detector PAN) costs nothing: producers write channel slices.
#pragma once
#include <stdint.h>

#define PF_PROGRAM_MAGIC 0x47504650  // "PFPG"
#define PF_PROGRAM_VERSION 11

enum PfElem : int32_t { PF_ELEM_ACT = 0, PF_ELEM_F32 = 1, PF_ELEM_I32 = 2, PF_ELEM_U8 = 3 };

struct PfHeader {
    int32_t magic, version, dtype, n_bufs, n_tensors, n_ops, const_bytes, arena_units_per_item;
    int32_t in_h, in_w, out_buf0, out_buf1, out_buf2, reserved0, reserved1, reserved2;
};
struct PfBufRec {
    int32_t etype, elems_per_item, offset_units, reserved;
};
struct PfTensorRec {
    int32_t buf, coff, ld, H, W, C, reserved0, reserved1;
};
#define PF_OP_FIELDS 39
struct PfOpRec {
    int32_t code;
    int32_t f[PF_OP_FIELDS];
};

enum PfOpCode : int32_t {
    PF_OP_STEM = 1,     // f: in_t(-1 = program input) out_t wt(u8 input, 1/255 folded) bias act wt(f32 input)
    PF_OP_CONV = 2,     // f: in_t out_t wt bias res_t gate_buf fbias_buf KH KW stride pad dil Cpad Npad N act
                        //    outCs amax_val_buf amax_idx_buf amaxN store_out cfg acc_scale(float bits) use_split
    PF_OP_DW = 3,       // f: in_t out_t wt bias K stride pad dil act
    PF_OP_UPCAT = 4,    // f: lo_t skip_t out_t
    PF_OP_GAP = 5,      // f: in_t out_buf
    PF_OP_FC = 6,       // f: x_buf y_buf wt bias K N act scale2 shift2 act2
    PF_OP_SCSE = 7,     // f: in_t out_t cse_buf ssew sse_b(float bits)
    PF_OP_HMDEC = 8,    // f: val_buf idx_buf feat_t offwt offbias P nslots loc_buf score_buf
    PF_OP_MAXPOOL = 9,  // f: in_t out_t            (2x2 stride 2, ceil mode)
    PF_OP_COPY = 10,    // f: in_t out_t out_cs up  (channel-strided copy, optional nearest x2 upsample)
    PF_OP_DETDEC = 11,  // f: in_t rows_buf row0 stride anchors(wt off, 6 floats) nrows_total
    PF_OP_ADDUP = 13,   // f: a_t b_t out_t shift act      out = act(a + nearest_up(b, 2^shift))
    PF_OP_MBCONV = 14,  // f: in_t out_t res_t w_exp b_exp w_dw b_dw w_pwl b_pwl K stride pad dil act MidPad KS CoutPad Cout Mid16 scale_exp scale_pwl (float bits) variant(0 split: KS = Cin/32, 1 exact f32: KS field = CinPad16, 2 no expand, 3 ShuffleV2 unit: + act_dw act_out out_cs pass_src_t pass_dst_t)
    PF_OP_EXPDW = 15,   // f: in_t out_t gap_buf w_exp b_exp w_dw b_dw K pad dil act Cpad Npad N acc_scale(float bits) stride(0 = 1)
    PF_OP_CHAIN = 16,   // f: in_t out_t n_convs C then n_convs x (wt bias acc_scale(float bits)): chain of BasicBlocks (two 3x3
                        //    convs + identity residual each), one face's map resident in LDS (k_chain.h); split programs only
    PF_OP_BLOCK = 17,   // f: in_t out_t C wt1 b1 s1 wt2 b2 s2 (s = acc_scale float bits): one BasicBlock, TR rows per workgroup, flat-K
                        //    weights (k_chain.h basic_block_kernel); split programs only
    PF_OP_DETUNIT = 18, // f: in_t out_t w1 b1 wd bd w2 b2 wd1 bd1 w3 b3 s1 s2 s3 (float bits) C K1 stride Cin: a whole ShuffleV2Block of the
                        //    detector per launch (k_det.h det_unit_kernel); in_t = the block's input (stride 1: both halves), out_t = its
                        //    2C-channel output (channel shuffle folded into the store); wd1 .. s3 = branch 1 of a stride-2 block; split programs only
    PF_OP_DETC3 = 19,   // f: srcA_t srcB_t(-1) out_t(-1) out2_t(-1) rows_buf(-1) wA bA wB bB wC bC wD bD wE bE anchors sA sB sC sD sE stride (float
                        //    bits) CIN tail upA row0 nrows_total: a C3 block of the detector's PAN head per launch (k_det.h det_c3_kernel) on the
                        //    concatenation [srcA (nearest x2 upsampled if upA) | srcB]; tail 1 = + a 1x1 conv (silu) into out2, tail 2 = + the
                        //    Detect conv (raw output into out2 if given) and its decode into rows_buf; split programs only
    PF_OP_DETSTEM = 20, // f: out_t w1_u8 w1_f32 b1 w2a b2a w2b b2b w3 b3 s1_u8 s1_f32 s2a s2b s3 (float bits): the detector's StemBlock (stem_1 3x3 s2,
                        //    stem_2a 1x1, stem_2b 3x3 s2, max-pool, stem_3 1x1) in one launch on the program input (k_det.h det_stem_kernel)
    // 21: PF_OP_LMFRONT (round 4: conv_stem + blocks.0.0 + blocks.1.0 in one launch) -- removed in round 6, see PF_OP_FRONT2
    PF_OP_HRB = 22,     // f: in_t out_t w1 b1 w2 b2 w3 b3 wd(-1) bd(-1) s1 s2 s3 sd (float bits) CIN: an HRNet Bottleneck (1x1 -> 3x3 -> 1x1 + shortcut, mid 64,
                        //    out 256; wd / bd = the first block's shortcut conv) in one launch (k_hrb.h hr_bottleneck_kernel); split programs only
    PF_OP_FUSEUP = 23,  // f: y_t out_t act nsrc then per source (<= 3): src_t wt bias shift: an HRNet fuse sum towards a higher-resolution branch,
                        //    out = act(y + sum_s nearest_up(conv1x1_s(src_s), 2^shift_s)), weights f32 [srcC][C padded to 4] (k_layers.h fuse_up_kernel);
                        //    f32 tensors only
    PF_OP_MBX = 24,     // f: in_t out_t(-1) res_t(-1) gap_buf(-1) gate_buf(-1) w1 ctile w2(-1) b2(-1) K pad dil act KS T Cout Cexp scale1 scale2 (float
                        //    bits) mode waves(8 | 16): a whole inverted-residual block at 16 x 16 with the face's input stationary in registers and
                        //    the expanded tile in LDS (k_mbx.h mbx_kernel); mode 0 = block without squeeze-excite, 1 = expand + depthwise -> per-face
                        //    channel means into gap_buf (the SE squeeze), 2 = expand + depthwise recomputed, x gate_buf, projected (+ res), 3 = mode 1
                        //    + the activated depthwise map stored in out_t (for the layer-wise gated projection); split programs only
    PF_OP_FC2 = 25,     // f: x_buf y_buf w1 b1(-1) K R act1 scale2(-1) shift2(-1) act1b w2 b2(-1) N act2 nparts(0 = 1) xscale(float bits): two dependent FCs on pooled vectors in one launch; x = xscale * sum of nparts partial vectors
                        //    (k_layers.h fc2_kernel: SE gate, cSE gate, ASPP pooled branch); K, R <= 960, R % 4 == 0, N % 4 == 0
    PF_OP_FRONT2 = 26,  // f: out_t w_stem_u8 w_stem_f32 b_stem s_u8 s_f32 (float bits) act_stem w_dw b_dw w_pw b_pw: conv_stem + blocks.0.0 of the Student encoder
                        //    (3x3 s2 3 -> 16 + act, depthwise 3x3 + relu -> 1x1 16 -> 16 + x) in one launch on the program input (k_front2.h); split programs only
    PF_OP_SEPUP = 12,   // f: lo_t skip_t out_t dwE(lo) dw_b(zeros: folded into pw_bias) pw_wt pw_bias Cpad Npad N act acc_scale(float bits) dw_w(skip) skipx_buf dw_w(lo, plain [9][C1]) dw_v(lo, [4 row classes][9][C1]: vertical interpolation folded in) gap_parts_buf + 1 (0 = none: per-tile channel sums of the output)
                        //    fused bilinear-x2-upsample + concat + depthwise 3x3 + pointwise conv (split kernels)
};

// tile configurations of conv_gemm_kernel (index = cfg field)
#define PF_CONV_NCFG 9
