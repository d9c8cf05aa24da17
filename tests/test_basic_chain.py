"""BasicBlock chains with the map resident in LDS (csrc/k_chain.h, PF_OP_CHAIN) against the same eight convs as separate
launches (conv3x3_halo_split_kernel / conv_gemm_split_kernel with a residual epilogue) in one program: HRNet's two
low-resolution branch shapes at a 256 x 256 crop, 72 channels @ 16 x 16 and 144 channels @ 8 x 8 (timm hrnet.py BasicBlock
x 4 per HighResolutionModule branch; TeacherNet, TRAIN/face_landmark/lib/core/base_trainer/model.py:306-311)."""
import numpy as np
import pytest

from peppa_pig_face_landmark_amd.graph import ir


def _program(c, hw, n_blocks, seed):
    rng = np.random.default_rng(seed)
    pb = ir.ProgramBuilder("f32s", 2 * hw, 2 * hw, keep_all=True)
    f0 = pb.stem(rng.normal(0, 0.6, (16, 3, 3, 3)), rng.normal(0, 0.1, 16), "relu")
    x = pb.conv(f0, rng.normal(0, 0.35, (c, 16, 1, 1)), rng.normal(0, 0.2, c), "none", out_name="x")
    std = np.sqrt(2.0 / (9 * c))
    blocks = [(rng.normal(0, std, (c, c, 3, 3)), rng.normal(0, 0.05, c), rng.normal(0, 0.6 * std, (c, c, 3, 3)), rng.normal(0, 0.05, c))
              for _ in range(n_blocks)]
    assert pb.basic_chain_supported(x, n_blocks)
    y = pb.basic_chain(x, blocks, out_name="chain.out")
    r = x
    for i, (w1, b1, w2, b2) in enumerate(blocks):
        m = pb.conv(r, w1, b1, "relu", pad=1)
        r = pb.conv(m, w2, b2, "relu", pad=1, res=r, out_name=f"ref.block{i}")
    loc, score = pb.buffer(196, ir.ELEM_F32, "loc"), pb.buffer(98, ir.ELEM_F32, "score")
    blob = pb.finish([loc, score])
    return blob, {"tensors": dict(pb.tensor_names)}, y, r


def _run(eng, c, hw, n_blocks, batch, seed):
    blob, info, _, _ = _program(c, hw, n_blocks, seed)
    eng.load_program(0, blob, batch)
    rng = np.random.default_rng(seed + 1)
    crops = rng.integers(0, 256, (batch, 2 * hw, 2 * hw, 3), dtype=np.uint8)
    eng.landmark_forward(crops)
    x = eng.read_tensor(0, info["tensors"]["x"], batch, (hw, hw, c))
    got = eng.read_tensor(0, info["tensors"]["chain.out"], batch, (hw, hw, c))
    ref = eng.read_tensor(0, info["tensors"][f"ref.block{n_blocks - 1}"], batch, (hw, hw, c))
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0.1 and (x < 0).any()      # the comparison is not vacuous
    assert (np.abs(ref).reshape(batch, -1).max(1) > 0.05).all()
    rel = np.abs(got - ref).max() / np.abs(ref).max()
    assert rel < 2e-5, (c, hw, n_blocks, rel)


@pytest.mark.parametrize("c,hw,n_blocks,batch", [(72, 16, 4, 2), (144, 8, 4, 3), (72, 16, 1, 1), (144, 8, 2, 1)])
def test_chain_equals_separate_convs_emu(emu_engine, c, hw, n_blocks, batch):
    _run(emu_engine, c, hw, n_blocks, batch, seed=100 + c + n_blocks)


@pytest.mark.gpu
@pytest.mark.parametrize("c,hw,n_blocks,batch", [(72, 16, 4, 5), (144, 8, 4, 7), (72, 16, 2, 300), (144, 8, 1, 1)])
def test_chain_equals_separate_convs_gpu(gpu_engine, c, hw, n_blocks, batch):
    _run(gpu_engine, c, hw, n_blocks, batch, seed=200 + c + n_blocks)


def _narrow_conv_case(eng, c, hw, batch, seed):
    """HRNet's narrow 3x3 convs (conv3x3_halo_split_kernel<32 | 48, 8, 1, 256> and layer1's <64, 4, 2, 256>: 256-pixel tiles)
    against torch conv2d on the tensor the engine itself produced upstream."""
    import torch
    rng = np.random.default_rng(seed)
    pb = ir.ProgramBuilder("f32s", 2 * hw, 2 * hw, keep_all=True)
    f0 = pb.stem(rng.normal(0, 0.6, (16, 3, 3, 3)), rng.normal(0, 0.1, 16), "relu")
    x = pb.conv(f0, rng.normal(0, 0.35, (c, 16, 1, 1)), rng.normal(0, 0.2, c), "none", out_name="x")
    w1, b1 = rng.normal(0, np.sqrt(2.0 / (9 * c)), (c, c, 3, 3)), rng.normal(0, 0.05, c)
    y = pb.conv(x, w1, b1, "relu", pad=1, res=x, out_name="y")
    blob = pb.finish([pb.buffer(196, ir.ELEM_F32, "loc"), pb.buffer(98, ir.ELEM_F32, "score")])
    eng.load_program(0, blob, batch)
    eng.landmark_forward(rng.integers(0, 256, (batch, 2 * hw, 2 * hw, 3), dtype=np.uint8))
    cp = (c + 3) // 4 * 4
    xv = eng.read_tensor(0, pb.tensor_names["x"], batch, (hw, hw, cp))[..., :c]
    got = eng.read_tensor(0, pb.tensor_names["y"], batch, (hw, hw, cp))
    assert not got[..., c:].any()                       # vector padding channels are written as zeros
    xt = torch.from_numpy(xv.astype(np.float64)).permute(0, 3, 1, 2)
    ref = torch.relu(torch.nn.functional.conv2d(xt, torch.from_numpy(w1), torch.from_numpy(b1), padding=1) + xt)
    ref = ref.permute(0, 2, 3, 1).numpy()
    rel = np.abs(got[..., :c] - ref).max() / np.abs(ref).max()
    assert rel < 5e-6, (c, hw, rel)          # f32 accumulation over up to 576 products


@pytest.mark.parametrize("c,hw,batch", [(18, 16, 3), (18, 32, 1), (36, 16, 2), (64, 64, 1), (128, 64, 1)])      # (128, 64): csrc/k_hero.h
def test_narrow_halo_convs_emu(emu_engine, c, hw, batch):
    _narrow_conv_case(emu_engine, c, hw, batch, seed=300 + c + hw)


@pytest.mark.gpu
@pytest.mark.parametrize("c,hw,batch", [(18, 64, 5), (36, 32, 9), (18, 32, 3), (36, 16, 4), (72, 16, 2), (64, 64, 3), (128, 64, 5)])
def test_narrow_halo_convs_gpu(gpu_engine, c, hw, batch):
    _narrow_conv_case(gpu_engine, c, hw, batch, seed=400 + c + hw)


def _block_case(eng, c, hw, batch, n_blocks, seed):
    """PF_OP_BLOCK (both convs of a BasicBlock in one launch, flat-K weights, one-row halo recompute) against the two separate
    conv launches, chained n_blocks deep; rows at the top / bottom image border and the ragged last conv1 tile round included."""
    rng = np.random.default_rng(seed)
    pb = ir.ProgramBuilder("f32s", 2 * hw, 2 * hw, keep_all=True)
    f0 = pb.stem(rng.normal(0, 0.6, (16, 3, 3, 3)), rng.normal(0, 0.1, 16), "relu")
    x = pb.conv(f0, rng.normal(0, 0.35, (c, 16, 1, 1)), rng.normal(0, 0.2, c), "none", out_name="x")
    std = np.sqrt(2.0 / (9 * c))
    y = r = x
    for i in range(n_blocks):
        w1, b1, w2, b2 = rng.normal(0, std, (c, c, 3, 3)), rng.normal(0, 0.05, c), rng.normal(0, 0.6 * std, (c, c, 3, 3)), rng.normal(0, 0.05, c)
        assert pb.basic_block_supported(y)
        y = pb.basic_block(y, w1, b1, w2, b2, out_name=f"fused.block{i}")
        m = pb.conv(r, w1, b1, "relu", pad=1)
        r = pb.conv(m, w2, b2, "relu", pad=1, res=r, out_name=f"ref.block{i}")
    blob = pb.finish([pb.buffer(196, ir.ELEM_F32, "loc"), pb.buffer(98, ir.ELEM_F32, "score")])
    eng.load_program(0, blob, batch)
    eng.landmark_forward(rng.integers(0, 256, (batch, 2 * hw, 2 * hw, 3), dtype=np.uint8))
    cp = (c + 3) // 4 * 4
    for i in range(n_blocks):
        got = eng.read_tensor(0, pb.tensor_names[f"fused.block{i}"], batch, (hw, hw, cp))
        ref = eng.read_tensor(0, pb.tensor_names[f"ref.block{i}"], batch, (hw, hw, cp))
        assert np.abs(ref).max() > 0.1 and not got[..., c:].any()
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        assert rel < 1e-5, (c, hw, i, rel)


@pytest.mark.parametrize("c,hw,batch,n_blocks", [(18, 16, 3, 2), (18, 64, 1, 1), (36, 32, 1, 2)])
def test_block_equals_separate_convs_emu(emu_engine, c, hw, batch, n_blocks):
    _block_case(emu_engine, c, hw, batch, n_blocks, seed=500 + c + hw)


@pytest.mark.gpu
@pytest.mark.parametrize("c,hw,batch,n_blocks", [(18, 64, 5, 4), (36, 32, 7, 4), (18, 16, 9, 2), (18, 64, 130, 1)])
def test_block_equals_separate_convs_gpu(gpu_engine, c, hw, batch, n_blocks):
    _block_case(gpu_engine, c, hw, batch, n_blocks, seed=600 + c + hw)


def _bottleneck_case(eng, h, w, batch, seed):
    """PF_OP_HRB (csrc/k_hrb.h: an HRNet Bottleneck in one launch; timm hrnet.py Bottleneck, TeacherNet layer1,
    TRAIN/face_landmark/lib/core/base_trainer/model.py:306-311) against the same convs as separate launches: the first block
    (64 channels in, 1x1 shortcut conv) followed by an identity-shortcut block, on a rectangular map whose 8 x 16 tiles are ragged
    in both directions (border tiles, halo pixels outside the image, tiles_x > 1)."""
    rng = np.random.default_rng(seed)
    pb = ir.ProgramBuilder("f32s", 2 * h, 2 * w, keep_all=True)
    f0 = pb.stem(rng.normal(0, 0.6, (16, 3, 3, 3)), rng.normal(0, 0.1, 16), "relu")
    x = pb.conv(f0, rng.normal(0, 0.35, (64, 16, 1, 1)), rng.normal(0, 0.2, 64), "relu", out_name="x")
    y = r = x
    cin = 64
    for i in range(2):
        w1, b1 = rng.normal(0, np.sqrt(2.0 / cin), (64, cin, 1, 1)), rng.normal(0, 0.05, 64)
        w2, b2 = rng.normal(0, np.sqrt(2.0 / 576), (64, 64, 3, 3)), rng.normal(0, 0.05, 64)
        w3, b3 = rng.normal(0, 0.5 * np.sqrt(2.0 / 64), (256, 64, 1, 1)), rng.normal(0, 0.05, 256)
        wd, bd = (rng.normal(0, np.sqrt(1.0 / cin), (256, cin, 1, 1)), rng.normal(0, 0.05, 256)) if i == 0 else (None, None)
        assert pb.hr_bottleneck_supported(y, 64, 256, wd is not None)
        y = pb.hr_bottleneck(y, w1, b1, w2, b2, w3, b3, wd, bd, out_name=f"fused.block{i}")
        m = pb.conv(r, w1, b1, "relu")
        m = pb.conv(m, w2, b2, "relu", pad=1)
        sc = pb.conv(r, wd, bd, "none") if wd is not None else r
        r = pb.conv(m, w3, b3, "relu", res=sc, out_name=f"ref.block{i}")
        cin = 256
    blob = pb.finish([pb.buffer(196, ir.ELEM_F32, "loc"), pb.buffer(98, ir.ELEM_F32, "score")])
    eng.load_program(0, blob, batch)
    eng.landmark_forward(rng.integers(0, 256, (batch, 2 * h, 2 * w, 3), dtype=np.uint8))
    for i in range(2):
        got = eng.read_tensor(0, pb.tensor_names[f"fused.block{i}"], batch, (h, w, 256))
        ref = eng.read_tensor(0, pb.tensor_names[f"ref.block{i}"], batch, (h, w, 256))
        assert np.isfinite(ref).all() and np.abs(ref).max() > 0.1
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        assert rel < 1e-5, (h, w, i, rel)


@pytest.mark.parametrize("h,w,batch", [(20, 40, 2), (8, 8, 1)])
def test_bottleneck_equals_separate_convs_emu(emu_engine, h, w, batch):
    _bottleneck_case(emu_engine, h, w, batch, seed=700 + h + w)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,batch", [(64, 64, 5), (20, 40, 9), (36, 24, 3)])
def test_bottleneck_equals_separate_convs_gpu(gpu_engine, h, w, batch):
    _bottleneck_case(gpu_engine, h, w, batch, seed=800 + h + w)


def _fuse_up_case(eng, h, w, c, src_c, batch, seed):
    """PF_OP_FUSEUP (csrc/k_layers.h fuse_up_kernel: every upsampled term of an HRNet fuse sum in one launch; timm hrnet.py
    HighResolutionModule fuse_layers[i][j > i], TeacherNet model.py:306-311) against the launches it replaces in the same program:
    1x1 conv at low resolution -> nearest upsample -> add, term by term, relu on the last."""
    rng = np.random.default_rng(seed)
    pb = ir.ProgramBuilder("f32s", 2 * h, 2 * w, keep_all=True)
    f0 = pb.stem(rng.normal(0, 0.6, (16, 3, 3, 3)), rng.normal(0, 0.1, 16), "relu")
    y = pb.conv(f0, rng.normal(0, 0.35, (c, 16, 1, 1)), rng.normal(0, 0.2, c), "none", out_name="y")
    srcs, t = [], f0
    for s, k in enumerate(src_c):                        # the lower branches: strided convs of the stem map
        t = pb.conv(t, rng.normal(0, np.sqrt(2.0 / (9 * pb.tensors[t].real_c)), (k, pb.tensors[t].real_c, 3, 3)), rng.normal(0, 0.1, k), "relu",
                    stride=2, pad=1)
        srcs.append(t)
    terms = [(srcs[s], rng.normal(0, np.sqrt(1.0 / k), (c, k, 1, 1)), rng.normal(0, 0.1, c), s + 1) for s, k in enumerate(src_c)]
    assert pb.fuse_up_supported(y, [(t, sh) for t, _, _, sh in terms])
    fused = pb.fuse_up(y, terms, "relu", out_name="fused")
    r = y
    for i, (t, wt, b, sh) in enumerate(terms):
        u = pb.conv(t, wt, b, "none")
        r = pb.add_up(r, u, sh, "relu" if i == len(terms) - 1 else "none", out_name="ref" if i == len(terms) - 1 else "")
    blob = pb.finish([pb.buffer(196, ir.ELEM_F32, "loc"), pb.buffer(98, ir.ELEM_F32, "score")])
    eng.load_program(0, blob, batch)
    eng.landmark_forward(rng.integers(0, 256, (batch, 2 * h, 2 * w, 3), dtype=np.uint8))
    cp = (c + 3) // 4 * 4
    got = eng.read_tensor(0, pb.tensor_names["fused"], batch, (h, w, cp))
    ref = eng.read_tensor(0, pb.tensor_names["ref"], batch, (h, w, cp))
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0.1 and (ref == 0).any() and not got[..., c:].any()
    rel = np.abs(got - ref).max() / np.abs(ref).max()
    assert rel < 2e-5, (h, w, c, rel)          # the reference's 1x1 convs are split-precision products, the fused ones plain f32 FMAs


@pytest.mark.parametrize("h,w,c,src_c,batch", [(32, 32, 18, (36, 72, 144), 2), (24, 40, 36, (72,), 1), (8, 8, 72, (144,), 3)])
def test_fuse_up_equals_separate_launches_emu(emu_engine, h, w, c, src_c, batch):
    _fuse_up_case(emu_engine, h, w, c, src_c, batch, seed=900 + h + c)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,c,src_c,batch", [(64, 64, 18, (36, 72, 144), 5), (32, 32, 36, (72, 144), 7), (16, 16, 72, (144,), 9), (24, 40, 18, (36,), 3)])
def test_fuse_up_equals_separate_launches_gpu(gpu_engine, h, w, c, src_c, batch):
    _fuse_up_case(gpu_engine, h, w, c, src_c, batch, seed=1000 + h + c)
