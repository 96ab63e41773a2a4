"""Pin the CPU oracle to the golden vectors captured from the reference.

Every fixture here was produced by tests/golden/make_golden.py running
/root/reference (cliora/net/{diora,cliora,trainer}.py, analysis/cky.py).
"""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden, params_from_golden
from oracle import chart_layout as CL
from oracle import diora_ref as R
from oracle import synth

TOL = 2e-6   # oracle vs reference, same torch build, CPU fp32


def close(a, b, tol=TOL, what=''):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * scale, '%s: max err %.3e (scale %.3e)' % (what, err, scale)


# ---------------------------------------------------------------- index tables
@pytest.mark.parametrize('L', list(range(2, 41)))
def test_index_tables_match_reference(golden, L):
    g = golden('index_tables.npz')
    assert np.array_equal(CL.level_offsets(L), g['off_%d' % L])
    li = np.concatenate([CL.inside_pairs(L, lv)[0] for lv in range(1, L)])
    ri = np.concatenate([CL.inside_pairs(L, lv)[1] for lv in range(1, L)])
    pi = np.concatenate([CL.outside_pairs(L, lv)[0] for lv in range(L - 2, -1, -1)])
    si = np.concatenate([CL.outside_pairs(L, lv)[1] for lv in range(L - 2, -1, -1)])
    assert np.array_equal(li, g['lidx_%d' % L])
    assert np.array_equal(ri, g['ridx_%d' % L])
    assert np.array_equal(pi, g['pidx_%d' % L])
    assert np.array_equal(si, g['sidx_%d' % L])
    assert len(li) == CL.n_inside_pairs(L) and len(pi) == CL.n_outside_pairs(L)


def test_survey_probe_L4():
    # L=4, level=1 -> par [9,8,8,7,7,9], sis [6,3,1,2,0,4]
    p, s = CL.outside_pairs(4, 1)
    assert p.tolist() == [9, 8, 8, 7, 7, 9] and s.tolist() == [6, 3, 1, 2, 0, 4]


# ---------------------------------------------------------------- DIORA charts + grads
def _run_diora(P, x, cot, share, normalize):
    for v in P.values():
        v.requires_grad_(True)
    x = x.clone().requires_grad_(True)
    out = R.diora_forward(P, x, x, share=share, normalize=normalize, training=True, keep_pairs=True)
    loss = sum((out[k] * v).sum() for k, v in cot.items())
    loss.backward()
    return out, x


@pytest.mark.parametrize('name', ['diora_c1.npz', 'diora_noshare.npz', 'diora_nonorm.npz', 'diora_len2.npz', 'diora_compress.npz'])
def test_diora_full_cases(golden, name):
    g = golden(name)
    m = g['meta']
    P = params_from_golden(g)
    if m['share']:   # the reference state_dict lists the aliased outside_* twins; drop them
        P = {k: v for k, v in P.items() if not k.startswith('outside_')}
    # the seeded synthesiser reproduces the stored parameters and inputs bit for bit
    P2, x2, cot2 = synth.diora_case(m['D'], m['B'], m['L'], m['seed'], share=m['share'], compress=m.get('compress', False))
    for k in P2:
        assert torch.equal(P2[k], P[k]), k
    assert np.array_equal(x2.numpy(), g['x_span'])
    cot = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('cot__')}
    for k in cot:
        assert torch.equal(cot2[k], cot[k])
    out, x = _run_diora(P, torch.from_numpy(g['x_span']), cot, m['share'], m['normalize'])
    for k in ('inside_h', 'inside_c', 'inside_s', 'outside_h', 'outside_c', 'outside_s'):
        close(out[k], g[k], what=k)
    L = m['L']
    for level in range(1, L):
        s = out['pair_s_in'][level]
        hook = (s - s.max(2, keepdim=True)[0])
        close(hook, g['hook_s_%d' % level], what='hook %d' % level)
    for k, v in g.items():
        if k.startswith('grad__') and k != 'grad__x_span':
            name_ = k[6:].replace('__', '.')
            close(P[name_].grad, v, tol=2e-5, what=k)
    close(x.grad, g['grad__x_span'], tol=2e-5, what='grad x')
    trees = R.cky_trees(out['pair_s_in'], m['B'], L)
    assert [str(t) for t in trees] == m['trees']
    assert [[list(s) for s in R.tree_spans(t)] for t in trees] == m['spans']


def test_diora_c2_shape_from_seed(golden):
    g = golden('diora_c2_small.npz')
    m = g['meta']
    assert m['torch'] == torch.__version__, 'fixture is tied to the torch build that drew the seeds'
    P, x, cot = synth.diora_case(m['D'], m['B'], m['L'], m['seed'])
    out, xg = _run_diora(P, x, cot, True, 'unit')
    cells = g['cells']
    for k in ('inside_h', 'inside_c', 'outside_h', 'outside_c'):
        close(out[k][:, cells], g[k + '__cells'], what=k)
        v = out[k].detach().numpy().astype(np.float64)
        assert abs(v.sum() - g[k + '__sum']) <= 1e-4 * max(1.0, g[k + '__abssum'])
    close(out['inside_s'], g['inside_s'], tol=5e-6, what='inside_s')
    close(out['outside_s'], g['outside_s'], tol=5e-6, what='outside_s')
    for k in g:
        if k.startswith('grad__') and k.endswith('__head'):
            name_ = k[6:-6].replace('__', '.')
            t = xg.grad if name_ == 'x_span' else P[name_].grad
            ref = g[k]
            close(t.reshape(-1)[:64], ref, tol=1e-4, what=k)
    trees = R.cky_trees(out['pair_s_in'], m['B'], m['L'])
    assert [str(t) for t in trees] == m['trees']


# ---------------------------------------------------------------- CLIORA
def _cliora_inputs(g, rg=False):
    t = {k: torch.from_numpy(g[k].copy()) for k in ('x_span', 'x_word', 'obj_span', 'obj_word')}
    if rg:
        for v in t.values():
            v.requires_grad_(True)
    return t


def test_cliora_eval(golden):
    g = golden('cliora_small.npz')
    m = g['meta']
    P = {k: v for k, v in params_from_golden(g).items() if not k.startswith('outside_')}
    t = _cliora_inputs(g)
    with torch.no_grad():
        out = R.diora_forward(P, t['x_span'], t['x_word'], t['obj_span'], t['obj_word'],
                              training=False, keep_pairs=True)
    for k in ('inside_h', 'inside_c', 'inside_s', 'outside_h', 'outside_c', 'outside_s',
              'all_atten_score', 'vg_atten_score', 'atten_score'):
        close(out[k], g['eval__' + k], what=k)
    trees = R.cky_trees(out['pair_s_in'], m['B'], m['L'])
    assert [str(t_) for t_ in trees] == m['trees']


def test_cliora_train_same_rng_and_grads(golden):
    g = golden('cliora_small.npz')
    m = g['meta']
    P = {k: v for k, v in params_from_golden(g, requires_grad=True).items() if not k.startswith('outside_')}
    t = _cliora_inputs(g, rg=True)
    torch.manual_seed(m['seed'] + 2)      # same RNG stream as the reference's nn.Dropout draws
    out = R.diora_forward(P, t['x_span'], t['x_word'], t['obj_span'], t['obj_word'], training=True)
    for k in ('inside_h', 'inside_s', 'outside_h', 'outside_s', 'all_atten_score', 'vg_atten_score', 'atten_score'):
        close(out[k], g['train__' + k], what=k)
    lc = R.contrastive_loss(out['inside_s'], out['outside_s'], out['all_atten_score'], 0.2, 1.0)
    lv = R.vg_loss(out['vg_atten_score'], 1.0)
    close(lc, g['train__contrastive_loss'], tol=1e-5, what='contrastive')
    close(lv, g['train__vg_loss'], tol=1e-5, what='vg')
    cot = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('cot__')}
    (lc + lv + sum((out[k] * v).sum() for k, v in cot.items())).backward()
    for k, v in g.items():
        if k.startswith('grad__'):
            name_ = k[6:].replace('__', '.')
            tt = t[name_].grad if name_ in t else P[name_].grad
            close(tt, v, tol=5e-5, what=k)


# ---------------------------------------------------------------- whole Net step
def _net_forward(g, P, vl):
    sent = torch.from_numpy(g['sentences'])
    neg = torch.from_numpy(g['neg_samples'])
    xs, xw = R.embed_forward(P['embed.embeddings.weight'], P['embed.mat'], P['embed.mat1'], sent)
    os_ = ow_ = None
    if vl:
        os_, ow_ = R.image_encoder_forward(P['img_encoder.fc.weight'], P['img_encoder.fc.bias'],
                                           P['img_encoder.fc_vis.weight'], P['img_encoder.fc_vis.bias'],
                                           torch.from_numpy(g['obj_feats']))
    D = {k[len('diora.'):]: v for k, v in P.items() if k.startswith('diora.') and not k.startswith('diora.outside_')}
    out = R.diora_forward(D, xs, xw, os_, ow_, training=False)
    losses = [R.reconstruction_loss(P['embed.embeddings.weight'], P['reconstruct_softmax_loss.mat'],
                                    sent, neg, out['outside_h'])]
    if vl:   # order of loss_funcs in trainer.py:187-197: reconstruct, vg, contrastive
        losses.append(R.vg_loss(out['vg_atten_score'], 1.0))
        losses.append(R.contrastive_loss(out['inside_s'], out['outside_s'], out['all_atten_score'], 0.2, 1.0))
    return torch.stack(losses).view(1, -1)


@pytest.mark.parametrize('name,vl', [('net_diora.npz', False), ('net_cliora.npz', True)])
def test_net_losses_and_adam_steps(golden, name, vl):
    g = golden(name)
    m = g['meta']
    P = params_from_golden(g)
    P.pop('reconstruct_softmax_loss.embeddings.weight')   # alias of embed.embeddings.weight
    train_keys = [k for k in P if not (vl and k == 'embed.embeddings.weight') and not k.startswith('diora.outside_')]
    if not vl:
        train_keys = [k for k in train_keys if not k.startswith('img_encoder.')]
    for k in train_keys:
        P[k].requires_grad_(True)
    tl = _net_forward(g, P, vl)
    close(tl, g['total_loss'], tol=1e-5, what='total_loss')
    tl.mean(0).sum().backward()
    for k in train_keys:
        gk = 'grad__' + k.replace('.', '__')
        if gk in g:
            close(P[k].grad, g[gk], tol=5e-5, what=gk)
    # Trainer._step x3: zero_grad, backward, clip_grad_norm_(5.0), Adam(lr, (0.9,0.999), 1e-8)
    params = [P[k] for k in train_keys]
    opt = torch.optim.Adam(params, lr=m['lr'], betas=(0.9, 0.999), eps=1e-8)
    for step in range(3):
        opt.zero_grad()
        tl = _net_forward(g, P, vl).mean(0).sum()
        assert abs(tl.item() - g['step_losses'][step]) <= 1e-4 * max(1.0, abs(g['step_losses'][step]))
        tl.backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
    for k in train_keys:
        close(P[k], g['after3__' + k.replace('.', '__')], tol=2e-4, what='after3 ' + k)


@pytest.mark.parametrize('name', ['treelstm_recon.npz', 'treelstm_recon_noshare.npz'])
def test_treelstm_reconstruction_matches_commented_reference_code(golden, name):
    """TreeLSTM is PARITY-UNPINNED: the reference ships it only as commented-out text (vg.py:28-76).
    The fixture was made by executing that text against the live DioraBase (shared functions, and a second
    outside compose / score module as diora.py:462-464 builds them); the oracle's own restatement must reproduce it."""
    g = golden(name)
    m = g['meta']
    assert m['reconstruction']
    share = m.get('share', True)
    P = {k: v.requires_grad_(True) for k, v in params_from_golden(g).items() if not (share and k.startswith('outside_'))}
    x = torch.from_numpy(g['x_span']).requires_grad_(True)
    out = R.diora_forward(P, x, x, arch='treelstm', training=True, keep_pairs=True, share=share)
    keys = ('inside_h', 'inside_c', 'inside_s', 'outside_h', 'outside_c', 'outside_s')
    for k in keys:
        close(out[k], g[k], what=k)
    sum((out[k] * torch.from_numpy(g['cot__' + k])).sum() for k in keys).backward()
    for k, v in g.items():
        if k.startswith('grad__') and k != 'grad__x_span':
            close(P[k[6:].replace('__', '.')].grad, v, tol=2e-5, what=k)
    close(x.grad, g['grad__x_span'], tol=2e-5, what='grad x')
    assert [str(t) for t in R.cky_trees(out['pair_s_in'], m['B'], m['L'])] == m['trees']
