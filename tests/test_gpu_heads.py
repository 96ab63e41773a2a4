"""The callers either side of the chart on this library's kernels (cliora_amd/heads.py, csrc/api_heads.hip) against the torch
formulas of the oracle (oracle/diora_ref.py restates cliora/net/trainer.py:25-224 and net/utils.py:37-55): values and every
gradient, at sizes with and without padding (D a multiple of 16 or not, negatives / regions not a multiple of 16 / 64, and the
reference's embedding widths that are not multiples of 16: 300 = GloVe, 1324 = w2v + ELMo).
Tolerance: 1e-4 of the tensor's scale (fp32 products, other summation order than ATen)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, what, tol=1e-4):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    sc = max(1.0, float(b.abs().max()))
    assert float((a - b).abs().max()) <= tol * sc, (what, float((a - b).abs().max()), sc)


@pytest.mark.parametrize('B,L,V,E,D', [(5, 7, 50, 32, 48), (64, 20, 2000, 1024, 400), (3, 4, 20, 16, 50), (4, 6, 80, 300, 400), (2, 5, 40, 1324, 64)])
def test_embed_projection(B, L, V, E, D):
    from cliora_amd import heads
    from oracle import diora_ref as R
    g = torch.Generator().manual_seed(1)
    emb, mat, mat1 = torch.randn(V, E, generator=g), torch.randn(D, E, generator=g), torch.randn(D, E, generator=g)
    tok = torch.randint(0, V, (B, L), generator=g)
    cs, cw = torch.randn(B, L, D, generator=g), torch.randn(B, L, D, generator=g)
    ref = [t.clone().requires_grad_(True) for t in (emb, mat, mat1)]
    xs, xw = R.embed_forward(ref[0], ref[1], ref[2], tok)
    ((xs * cs).sum() + (xw * cw).sum()).backward()
    dev = [t.clone().cuda().requires_grad_(True) for t in (emb, mat, mat1)]
    idx = tok.cuda().reshape(-1)
    ys, yw = heads.proj(dev[0], idx, dev[1]).view(B, L, D), heads.proj(dev[0], idx, dev[2]).view(B, L, D)
    ((ys * cs.cuda()).sum() + (yw * cw.cuda()).sum()).backward()
    _close(ys, xs, 'x_span'); _close(yw, xw, 'x_word')
    for a, b, n in zip(dev, ref, ('embeddings', 'mat', 'mat1')):
        _close(a.grad, b.grad, n)


@pytest.mark.parametrize('B,R,K,D', [(4, 36, 2048, 400), (3, 5, 48, 50)])
def test_image_encoder_projection(B, R, K, D):
    from cliora_amd import heads
    from oracle import diora_ref as Rf
    g = torch.Generator().manual_seed(2)
    x = torch.relu(torch.randn(B, R, K, generator=g))
    ps = [0.05 * torch.randn(*s, generator=g) for s in ((D, K), (D,), (D, K), (D,))]
    c0, c1 = torch.randn(B, R, D, generator=g), torch.randn(B, R, D, generator=g)
    ref = [t.clone().requires_grad_(True) for t in ps]
    y0, y1 = Rf.image_encoder_forward(ref[0], ref[1], ref[2], ref[3], x)
    ((y0 * c0).sum() + (y1 * c1).sum()).backward()
    dev = [t.clone().cuda().requires_grad_(True) for t in ps]
    xd = x.cuda()
    z0 = heads.proj(xd, None, dev[0], dev[1]).view(B, R, D)
    z1 = heads.proj(xd, None, dev[2], dev[3]).view(B, R, D)
    ((z0 * c0.cuda()).sum() + (z1 * c1.cuda()).sum()).backward()
    _close(z0, y0, 'span'); _close(z1, y1, 'word')
    for a, b, n in zip(dev, ref, ('fc.weight', 'fc.bias', 'fc_vis.weight', 'fc_vis.bias')):
        _close(a.grad, b.grad, n)


@pytest.mark.parametrize('B,L,V,E,D,K', [(4, 6, 60, 32, 48, 7), (64, 20, 2000, 1024, 400, 100), (2, 5, 40, 16, 50, 20), (3, 7, 90, 300, 400, 25)])
def test_reconstruction_loss(B, L, V, E, D, K):
    from cliora_amd import heads
    from oracle import diora_ref as R
    g = torch.Generator().manual_seed(3)
    C = L * (L + 1) // 2
    emb, mat = torch.randn(V, E, generator=g), 0.2 * torch.randn(D, E, generator=g)
    oh = torch.nn.functional.normalize(torch.randn(B, C, D, generator=g), dim=-1)
    tok = torch.randint(0, V, (B, L), generator=g)
    neg = torch.randperm(V, generator=g)[:K]
    ref = [t.clone().requires_grad_(True) for t in (emb, mat, oh)]
    lr = R.reconstruction_loss(ref[0], ref[1], tok, neg, ref[2])
    (3.0 * lr).backward()
    dev = [t.clone().cuda().requires_grad_(True) for t in (emb, mat, oh)]
    ld = heads.recon_loss(dev[0], dev[1], dev[2], tok.cuda(), neg.cuda())
    (3.0 * ld).backward()
    assert abs(float(ld) - float(lr)) <= 1e-4 * max(1.0, abs(float(lr)))
    for a, b, n in zip(dev, ref, ('embeddings', 'mat', 'outside_h')):
        _close(a.grad, b.grad, n)


@pytest.mark.parametrize('B,L,R', [(5, 7, 36), (64, 20, 36), (3, 4, 70), (2, 3, 1)])
def test_vg_loss(B, L, R):
    from cliora_amd import heads
    from oracle import diora_ref as Rf
    g = torch.Generator().manual_seed(4)
    vg = torch.randn(B, B, L, R, generator=g)
    ref = vg.clone().requires_grad_(True)
    lr = Rf.vg_loss(ref, 0.7)
    (2.0 * lr).backward()
    dev = vg.clone().cuda().requires_grad_(True)
    ld = heads.vg_loss(dev, 0.7)
    (2.0 * ld).backward()
    assert abs(float(ld) - float(lr)) <= 1e-5 * max(1.0, abs(float(lr)))
    _close(dev.grad, ref.grad, 'vg_atten_score', 1e-5)


def test_fused_clip_adam_matches_torch():
    """clip_grad_norm_(5.0) + Adam over one flat buffer against torch's own, three steps, both the clipping and the non-clipping regime."""
    from cliora_amd import heads
    g = torch.Generator().manual_seed(5)
    shapes = [(40, 30), (17,), (5, 3, 4), (1,)]
    for scale in (0.01, 30.0):          # total norm well under / well over max_norm
        p_ref = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
        p_dev = [torch.nn.Parameter(p.detach().clone()) for p in p_ref]
        opt = torch.optim.Adam(p_ref, lr=2e-3, betas=(0.9, 0.999), eps=1e-8)
        fused = heads.FusedClipAdam(p_dev, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=5.0)
        for step in range(3):
            grads = [scale * torch.randn(*s, generator=g).cuda() for s in shapes]
            for p, q, gr in zip(p_ref, p_dev, grads):
                p.grad = gr.clone()
                q.grad = gr.clone()
            torch.nn.utils.clip_grad_norm_(p_ref, 5.0)
            opt.step()
            fused.step()
            for p, q in zip(p_ref, p_dev):
                assert float((p - q).abs().max()) <= 1e-6 * max(1.0, float(p.abs().max())), (scale, step)


def test_rows_scatter_add_matches_index_add_and_is_deterministic():
    """cliora_rows_scatter_add (the backward of F.embedding, trainer.py:219: the embedding table trains when emb = none) against
    torch's zeros + index_add_, with repeated ids (summed in ascending order, no atomics: the same bits every run), ids out of the
    batch untouched (zero), and the sizes of the reconstruction head's lookup (B L + k_neg rows of 1024)."""
    import ctypes as C
    from cliora_amd import _lib
    g = torch.Generator().manual_seed(3)
    for n, K, V in ((1380, 1024, 10000), (37, 48, 11), (1, 16, 5)):
        idx = torch.randint(0, V, (n,), generator=g)
        if n > 8:
            idx[5] = idx[2]; idx[n - 1] = idx[2]; idx[7] = idx[6]          # repeats, one of them three times
        rows = torch.randn(n, K, generator=g)
        want = torch.zeros(V, K).index_add_(0, idx, rows)
        outs = []
        rows_d, idx_d = rows.cuda(), idx.cuda()
        for rep in range(2):
            out = torch.full((V, K), float('nan'), device='cuda')          # the call clears the table itself
            rc = _lib.lib().cliora_rows_scatter_add(C.c_void_p(rows_d.data_ptr()), C.c_void_p(idx_d.data_ptr()), n, K,
                                                    C.c_void_p(out.data_ptr()), V, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(rc, 'cliora_rows_scatter_add')
            torch.cuda.synchronize()
            outs.append(out.cpu())
        assert torch.equal(outs[0], outs[1])
        assert float((outs[0] - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())), (n, K, V)


def test_rows_scatter_add_segments_is_the_sum_of_the_lookups():
    """cliora_rows_scatter_add_segments: the table gradient of a whole step -- the reconstruction loss's positives + negatives and Embed's
    lookups (trainer.py:54-58, :219) -- from ONE launch, against torch's zeros + index_add_ per lookup; tokens shared between the
    lookups, a token repeated more often than the kernel's in-LDS list holds (its ordered fallback), empty segments; same bits every run;
    a segment order that matters only to rounding is the given one (bitwise equal to the one-segment call on the concatenation)."""
    import ctypes as C
    from cliora_amd import _lib
    g = torch.Generator().manual_seed(5)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for sizes, K, V in (((1380, 1280), 1024, 10000), ((9, 0, 300, 4), 48, 7), ((3,), 16, 5), ((0, 0), 16, 5)):
        idx = [torch.randint(0, V, (n,), generator=g) for n in sizes]
        rows = [torch.randn(n, K, generator=g) for n in sizes]
        big = max(range(len(sizes)), key=lambda k: (k > 0, sizes[k])) if len(sizes) > 1 else 0
        if big and sizes[0] > 8 and sizes[big] >= 6:
            idx[big][:6] = idx[0][:6]                              # tokens that both lookups hit
        want = torch.zeros(V, K)
        for i, r in zip(idx, rows):
            want.index_add_(0, i, r)
        rows_d, idx_d = [r.cuda() for r in rows], [i.cuda() for i in idx]
        n = len(sizes)
        outs = []
        for rep in range(2):
            out = torch.full((V, K), float('nan'), device='cuda')
            rc = _lib.lib().cliora_rows_scatter_add_segments((C.c_void_p * n)(*[r.data_ptr() for r in rows_d]), (C.c_void_p * n)(*[i.data_ptr() for i in idx_d]),
                                                             (C.c_int32 * n)(*sizes), n, K, C.c_void_p(out.data_ptr()), V, st())
            _lib.check(rc, 'cliora_rows_scatter_add_segments')
            torch.cuda.synchronize()
            outs.append(out.cpu())
        assert torch.equal(outs[0], outs[1])
        assert float((outs[0] - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), (sizes, K, V)
        if sum(sizes):
            cat_r, cat_i = torch.cat(rows_d), torch.cat(idx_d)
            one = torch.empty((V, K), device='cuda')
            _lib.check(_lib.lib().cliora_rows_scatter_add(C.c_void_p(cat_r.data_ptr()), C.c_void_p(cat_i.data_ptr()), int(cat_i.numel()), K,
                                                          C.c_void_p(one.data_ptr()), V, st()), 'cliora_rows_scatter_add')
            assert torch.equal(one.cpu(), outs[0])
    rc = _lib.lib().cliora_rows_scatter_add_segments((C.c_void_p * 5)(), (C.c_void_p * 5)(), (C.c_int32 * 5)(), 5, 16, C.c_void_p(1), 5, st())
    assert rc != 0 and b'segments' in _lib.lib().cliora_last_error()


@pytest.mark.parametrize('vl', [False, True])
def test_deferred_table_gradient_is_the_autograd_one(vl):
    """harness.Trainer.step assembles the embedding table's gradient once per step from every lookup's rows (heads.DeferredTableGrads);
    with the switch off autograd scatters per lookup and adds.  Same losses, same parameters after three steps (to the rounding of the
    one sum whose order differs: a token that two lookups share), and a parameter that never receives a gradient (embed.mat1 in a
    text-only net) stays exactly where it was."""
    from cliora_amd import harness as H
    res = {}
    for defer in (True, False):
        torch.manual_seed(11)
        V, E, D, B, L, K = 300, 64, 48, 6, 7, 20
        emb = torch.nn.Embedding(V, E)
        net = H.build_net(D, emb, obj_feats=vl, img_dim=32, k_neg=K, vg_loss=vl, use_contr=vl).cuda()
        if vl:
            emb.weight.requires_grad = True                    # exercise the deferred path in the vision-language net too
            for p in net.img_encoder.parameters():
                torch.nn.init.normal_(p, std=0.05)
        tr = H.Trainer(net, lr=2e-3)
        tr.defer_table_grads = defer
        g = torch.Generator().manual_seed(12)
        bm = dict(sentences=torch.randint(0, 40, (B, L), generator=g).cuda(), neg_samples=torch.randperm(V, generator=g)[:K].cuda())
        bm['neg_samples'][:3] = bm['sentences'][0, :3]         # negatives that are also words of the batch
        if vl:
            bm['obj_feats'] = torch.randn(B, 36, 32, generator=g).cuda()
        net.train = lambda mode=True, net=net: torch.nn.Module.train(net, False)      # dropout off: the two runs must see the same step
        mat1_0 = net.embed.mat1.detach().clone()
        losses = [tr.step(bm, train=True)['total_loss'] for _ in range(3)]
        res[defer] = (losses, {k: p.detach().clone() for k, p in net.named_parameters()})
        if not vl:
            assert torch.equal(net.embed.mat1.detach(), mat1_0)
    for a, b in zip(res[True][0], res[False][0]):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b))
    for k in res[False][1]:
        a, b = res[True][1][k], res[False][1][k]
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), k
