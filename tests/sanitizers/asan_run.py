"""Driver of tests/sanitizers/run.sh (expects ASan/UBSan preloaded): golden traces and batched paths through the
sanitizer builds of the oracle and of the kernel-body harness."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.oracle as orc
ASAN_ORACLE = os.path.join(ROOT, "oracle", "liboracle_asan.so")
real_cdll = ctypes.CDLL
def fake(path, *a, **k):
    return real_cdll(ASAN_ORACLE if str(path).endswith("liboracle.so") else path, *a, **k)
orc.C.CDLL = fake
import __graft_entry__ as ge
pkg = ge.package()
H = os.path.join(ROOT, "tests", "cpu_harness", "libtetris_cpu_harness_asan.so")
from tests import replay, engines
# 1. golden traces through both sanitized builds (incl. get_actions and colours)
for name in ('greedy_2p','keys_2p','actions_2p','keys_2p_22','rt_2p_sz'):
    tr = replay.load_trace(name)
    replay.replay(tr, lambda P,Hh,W,pc,sd: orc.OracleBatch(1,P,Hh,W,pieces=pc,seeds=sd), fields=replay.VISIBLE+replay.HIDDEN+replay.COLOUR_ONLY, check_actions=True, max_events=900)
    replay.replay(tr, lambda P,Hh,W,pc,sd: pkg.TetrisBatch(1,P,Hh,W,pieces=pc,seeds=sd,lib_path=H,colours=True), fields=replay.VISIBLE+replay.HIDDEN+replay.COLOUR_ONLY, check_actions=True, max_events=900)
    print('trace', name, 'ok')
# 2. batched paths: rollout, enumerate, observe_packed, snapshot, split stages
n=200
for P in (1,2):
    seeds = orc.episode_seed(np.arange(n),0)
    e = pkg.TetrisBatch(n,P,20,10,seeds=seeds,lib_path=H); r = orc.OracleBatch(n,P,20,10,seeds=seeds)
    c1,_ = e.rollout_random(5,20); _,c2 = r.rollout_random(100, threads=2)
    assert c1.tolist()==c2.tolist()
    e.enumerate_drops(player=0); r.enumerate_drops(player=0); e.observe_packed(player=0); e.get_actions(player=0)
    b = e.snapshot(); e.restore(b); engines.assert_same_state(e, r, where='asan')
print('batched ok')
