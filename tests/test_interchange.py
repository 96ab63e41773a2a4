"""Checkpoint interchange and evaluation bookkeeping against files and values
produced by the reference (tests/golden/make_golden.py::interchange_case): Trainer.save_model / load_model
(trainer.py:383-435), get_actions / get_spans / get_stats (analysis/utils.py:3-64), replace_leaves / postprocess /
the F1 accumulation and the parse.jsonl record (scripts/parse.py:63-98, 215-234, 270-290).  CPU only."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN as GOLDEN_DIR, load_golden
from cliora_amd import harness as H
from cliora_amd import interchange as X


def _net(g, seed):
    m = g['meta']
    torch.manual_seed(seed)
    emb = torch.nn.Embedding(m['V'], 16)
    return H.build_net(m['D'], emb, obj_feats=False, img_dim=20, k_neg=m['K'])


def _state(net):
    return {k.replace('.', '__'): v.detach().cpu().numpy() for k, v in net.state_dict().items()}


def _want(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def test_reference_checkpoints_load_like_the_reference_loader():
    g = load_golden('interchange.npz')
    first = _want(g, 'dst0__')
    for tag, fname, origin_emb in (('noemb', 'ref_model_noemb.pt', False), ('emb', 'ref_model_emb.pt', True), ('ddp', 'ref_model_ddp.pt', True)):
        net = _net(g, 1)
        assert set(_state(net)) == set(first), 'parameter names differ from the reference Net'
        net.load_state_dict({k.replace('__', '.'): torch.from_numpy(v.copy()) for k, v in first.items()})   # the receiving net's own init
        taken, kept = X.load_model(origin_emb, net, os.path.join(GOLDEN_DIR, fname))
        want = _want(g, 'loaded_%s__' % tag)
        got = _state(net)
        for k in want:
            assert np.array_equal(got[k], want[k]), (tag, k)
        emb_keys = [k for k in got if 'embeddings' in k]
        assert emb_keys and all((k.replace('__', '.') in kept) == (not origin_emb) for k in emb_keys)


def test_checkpoints_written_here_have_the_reference_layout(tmp_path):
    g = load_golden('interchange.npz')
    net = _net(g, 2)
    src = _want(g, 'src__')
    net.load_state_dict({k.replace('__', '.'): torch.from_numpy(v.copy()) for k, v in src.items()})
    for save_emb, ref_file in ((False, 'ref_model_noemb.pt'), (True, 'ref_model_emb.pt')):
        path = str(tmp_path / ('m%d.pt' % save_emb))
        X.save_model(net, save_emb, path)
        mine = torch.load(path, map_location='cpu')
        ref = torch.load(os.path.join(GOLDEN_DIR, ref_file), map_location='cpu')
        assert list(mine) == list(ref) == ['state_dict']
        assert list(mine['state_dict']) == list(ref['state_dict'])            # same keys in the same order
        for k in ref['state_dict']:
            assert torch.equal(mine['state_dict'][k], ref['state_dict'][k]), k


def _tup(x):
    return tuple(_tup(y) for y in x) if isinstance(x, list) else x


def test_spans_f1_and_parse_records_match_the_reference():
    g = load_golden('interchange.npz')
    m = g['meta']
    f1 = X.SpanF1()
    for c in m['cases']:
        tree = _tup(c['tree'])
        assert X.tree_spans(tree) == [tuple(s) for s in c['spans']]
        assert X.flatten_tree(tree) == c['flat']
        words = X.replace_leaves(tree, c['tokens'])
        assert words == c['replaced']
        assert json.loads(json.dumps(X.postprocess(words, c['tokens']))) == c['post']
        pred = f1.add(tree, [tuple(s) for s in c['gold']])
        rec = json.loads(X.parse_record(7, tree, c['tokens'], gold_spans=[tuple(s) for s in c['gold']], pred_spans=sorted(pred), post=True))
        assert list(rec) == ['example_id', 'tree', 'tree_index_conll', 'sentence', 'gold_spans', 'pred_spans', 'pred_boxes']
        assert rec['example_id'] == '7' and rec['tree'] == c['post'] and rec['sentence'] == c['tokens']
        assert rec['tree_index_conll'] == c['tree']
    assert abs(f1.corpus_f1 - m['corpus_f1']) < 1e-12
    assert abs(f1.sentence_f1 - m['sent_f1']) < 1e-12


def test_loaded_checkpoint_reproduces_the_reference_run_on_the_oracle():
    """tests/golden/interchange_run.npz (the reference Net after ITS load_model, run on a fixture batch): the native loader's parameters
    fed to the CPU oracle give the reference's charts and loss -- the CPU half of row f4 (the GPU half: tests/test_gpu_interchange.py)."""
    from oracle import diora_ref as R
    g = load_golden('interchange_run.npz')
    m = g['meta']
    sent, neg = torch.from_numpy(g['sentences']), torch.from_numpy(g['neg_samples'])
    for tag, fname, origin_emb in (('noemb', 'ref_model_noemb.pt', False), ('ddp', 'ref_model_ddp.pt', True)):
        torch.manual_seed(3)
        net = H.build_net(m['D'], torch.nn.Embedding(m['V'], 16), obj_feats=False, img_dim=20, k_neg=m['K'])
        net.load_state_dict({k[len('dst0__'):].replace('__', '.'): torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith('dst0__')})
        X.load_model(origin_emb, net, os.path.join(GOLDEN_DIR, fname))
        sd = net.state_dict()
        P = {k[len('diora.'):]: v for k, v in sd.items() if k.startswith('diora.')}
        xs, xw = R.embed_forward(sd['embed.embeddings.weight'], sd['embed.mat'], sd['embed.mat1'], sent)
        ref = R.diora_forward(P, xs, xw)
        for k in ('inside_h', 'inside_s', 'outside_h', 'outside_s'):
            assert np.abs(ref[k].numpy() - g['%s__%s' % (tag, k)]).max() <= 2e-6 * max(1.0, np.abs(g['%s__%s' % (tag, k)]).max()), (tag, k)
        loss = R.reconstruction_loss(sd['embed.embeddings.weight'], sd['reconstruct_softmax_loss.mat'], sent, neg, ref['outside_h'])
        assert abs(float(loss) - float(g['%s__total_loss' % tag].reshape(-1)[0])) <= 2e-6 * max(1.0, abs(float(loss)))


def test_build_net_arch_switch():
    """trainer.py:518-526: 'mlp' picks the text / vision-language DioraMLP, anything else raises; this library adds 'treelstm' (config 5)."""
    import pytest
    emb = torch.nn.Embedding(11, 16)
    assert type(H.build_net(16, emb, arch='mlp').diora).__module__.endswith('.diora')
    assert type(H.build_net(16, torch.nn.Embedding(11, 16), obj_feats=True, img_dim=16, arch='mlp').diora).__module__.endswith('.cliora')
    net = H.build_net(16, emb, arch='treelstm')
    assert type(net.diora).__name__ == 'DioraTreeLSTM'
    assert {'diora.inside_compose_func.W', 'diora.inside_compose_func.U', 'diora.inside_compose_func.B', 'diora.inside_score_func.mat',
            'diora.root_vector_out_h', 'diora.root_vector_out_c'} <= set(net.state_dict())
    with pytest.raises(NotImplementedError):
        H.build_net(16, emb, arch='hard')
    with pytest.raises(NotImplementedError):
        H.build_net(16, emb, obj_feats=True, arch='treelstm')
