"""SURVEY.md section 8(b) "Selection": with cliora_amd/shim ahead of the reference on sys.path, the reference's own
build_net (cliora/net/trainer.py:518-526, 552) constructs the MI355X-native chart modules without any edit to the
reference.  Needs the reference checkout (this container only); construction is CPU-side, no chart call is made."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

SCRIPT = r'''
import sys, types, argparse
sys.modules['cv2'] = types.ModuleType('cv2')
import torch
import cliora.net.diora as d, cliora.net.cliora as c, cliora.net.trainer as t, cliora.net.utils as u
assert d.__file__.startswith(ROOT) and c.__file__.startswith(ROOT), (d.__file__, c.__file__)
assert t.__file__.startswith(REF) and u.__file__.startswith(REF), (t.__file__, u.__file__)
for obj_feats in (False, True):
    opt = argparse.Namespace(lr=1e-3, hidden_dim=16, k_neg=3, normalize='unit', cuda=False, local_rank=0, share=True, arch='mlp',
                             obj_feats=obj_feats, multigpu=False, emb='none', margin=1.0, vl_margin=0.2, hinge_margin=1.0,
                             alpha_contr=1.0, vg_loss=obj_feats, alpha_vg=1.0, use_contr=obj_feats, use_contr_ce=False,
                             visualize=False, load_model_path=None, experiment_name='shim-test')
    trainer = t.build_net(opt, torch.nn.Embedding(20, 8))
    diora = trainer.net.diora
    want = 'cliora_amd.cliora' if obj_feats else 'cliora_amd.diora'
    assert type(diora).__module__ == want, type(diora).__module__
    # same state_dict keys as the reference's own module built with the same arguments (loaded from its file under another name)
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_mod', REF + '/cliora/net/' + ('cliora.py' if obj_feats else 'diora.py'))
    ref_mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_mod)
    want_keys = sorted(ref_mod.DioraMLP(16, outside=True, normalize='unit', compress=False, share=True).state_dict())
    assert sorted(diora.state_dict()) == want_keys, (sorted(diora.state_dict()), want_keys)
print('shim ok')
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'cliora')), reason='needs the reference checkout')
def test_reference_build_net_selects_the_native_modules():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'cliora_amd', 'shim'), ROOT, REF]))
    code = 'ROOT = %r\nREF = %r\n' % (ROOT, REF) + SCRIPT
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd='/tmp', capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'shim ok' in r.stdout, r.stdout + r.stderr
