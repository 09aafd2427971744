"""Sensitivity check of tests/test_gpu_finetune_loop.py::test_pipelined_uploads_wait_for_pending_work_on_the_callers_stream: the same test
body with the ordering of the side-stream fills switched OFF (sampling._ORDER_SIDE_FILLS = False, the state before round 6).  If the
hazard of ADVICE round 5 is real on this box the unordered run produces NaN / wrong poses at least sometimes; the ordered run never.
    python tools/check_upload_ordering.py [repeats]
Round 6, MI355X: {'ordered': '0 of 6 runs corrupted', 'unordered': '0 of 6 runs corrupted'} -- the hazard is masked today by the internal
synchronisations of cbd_set_complex (wave 0) and cbd_sample* (stream sync after the sigma upload); the ordering does not rely on them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import confidence_bootstrapping_amd.sampling as sp
import test_gpu_finetune_loop as t

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = {}
for ordered in (True, False):
    sp._ORDER_SIDE_FILLS = ordered
    bad = 0
    for _ in range(n):
        try:
            t.test_pipelined_uploads_wait_for_pending_work_on_the_callers_stream()
        except AssertionError:
            bad += 1
    res["ordered" if ordered else "unordered"] = f"{bad} of {n} runs corrupted"
sp._ORDER_SIDE_FILLS = True
print(res)
