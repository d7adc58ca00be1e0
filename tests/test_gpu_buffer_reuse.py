"""GPU: the spare-buffer slot (mi_set_global_option("keep_buffers")).  A stateless matching_<method> call prepares a gallery, searches
and destroys it (src/utils/nnsearch.py:687-706 re-normalises the gallery inside every call); the buffers of the destroyed gallery
are handed to the next one of the same sizes.  Nothing a search reads may depend on what the previous owner left there."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _session_mode():
    """The mode the session runs in (tests/conftest.py: ISEHR_KEEP_BUFFERS=0 switches the spare slots off for the WHOLE suite;
    ADVICE r05: these tests used to leave them on for every test behind them)."""
    return int(os.environ.get("ISEHR_KEEP_BUFFERS", "1"))


def _snapshot(g, q, k, n):
    idx, sc, _ = g.search(q, k)
    return g.get_rows(0, n).tobytes(), tuple(g.norm_bounds()), idx.tobytes(), sc.tobytes()


@pytest.mark.parametrize("n,d", [(5000, 2048), (4099, 2048), (777, 100)])
def test_recycled_buffers_give_the_same_gallery_as_fresh_ones(n, d):
    import torch
    from isehr_amd import _lib
    from isehr_amd.synth import synth_rows
    from oracle import retrieval_oracle as oracle
    a = synth_rows(11, 0, n, d) * 3.0                 # the previous owner: other rows, larger norms
    b = synth_rows(12, 0, n, d)
    q = synth_rows(13, 0, 6, d)
    k = 10
    _lib.set_global_option("keep_buffers", 0)
    try:
        g = _lib.Gallery.from_host(b)
        fresh = _snapshot(g, q, k, n)
        g.close()
        _lib.set_global_option("keep_buffers", 1)
        g = _lib.Gallery.from_host(a)
        g.search(q, k)
        g.close()                                     # -> the spare slot
        g = _lib.Gallery.from_host(b)                 # same sizes: takes the spare
        again = _snapshot(g, q, k, n)
        g.close()
        assert again == fresh
        # an appendable gallery of the same capacity, filled only partly: the rows beyond are the previous owner's bytes
        m = n - n // 3
        t = torch.from_numpy(b).cuda()
        g = _lib.Gallery.empty(n, d)
        g.append_device(t.data_ptr(), m, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        idx, sc, _ = g.search(q, k)
        g.close()
        bn = b[:m] / np.linalg.norm(b[:m].astype(np.float64), axis=1, keepdims=True)
        qn = q / np.linalg.norm(q.astype(np.float64), axis=1, keepdims=True)
        assert oracle.check_topk_parity(idx, (qn @ bn.T), k, 1e-6) == []
    finally:
        _lib.set_global_option("keep_buffers", _session_mode())


def test_keep_buffers_zero_frees_the_spare():
    import torch
    from isehr_amd import _lib
    from isehr_amd.synth import synth_rows
    rows = synth_rows(21, 0, 20000, 2048)
    torch.cuda.synchronize()
    _lib.set_global_option("keep_buffers", 1)
    g = _lib.Gallery.from_host(rows)
    g.close()
    try:
        free_kept = torch.cuda.mem_get_info()[0]
        held = _lib.get_global_option("spare_bytes")      # what a co-tenant can ask for (mi_get_global_option)
        assert held >= 20000 * 2048 * 6 and _lib.get_global_option("keep_buffers") == 1
        _lib.set_global_option("keep_buffers", 0)         # frees the slot
        free_after = torch.cuda.mem_get_info()[0]
        assert free_after - free_kept >= 20000 * 2048 * 6 * 0.9 and _lib.get_global_option("spare_bytes") == 0
        # "release_spares": the slots are emptied, the mode stays
        _lib.set_global_option("keep_buffers", 1)
        g = _lib.Gallery.from_host(rows)
        g.close()
        assert _lib.get_global_option("spare_bytes") > 0
        _lib.set_global_option("release_spares", 1)
        assert _lib.get_global_option("spare_bytes") == 0 and _lib.get_global_option("keep_buffers") == 1
        with pytest.raises(RuntimeError):
            _lib.set_global_option("keep_buffers", 2)
    finally:
        _lib.set_global_option("keep_buffers", _session_mode())


def test_workspace_is_not_recycled_under_a_search_in_flight():
    """Commit 87f05b2 closed this by inspection (VERDICT r05 Weak #7): a workspace parked in the spare slot could still be used
    by launches on a caller's non-blocking stream when the next handle took and cleared it.  tests/_hazard.py is the scenario;
    scripts/recycle_hazard_ab.sh runs the same function against a build without the synchronisation in ws_free, where it
    fails (profiles/r06_recycle_hazard_ab.txt)."""
    from isehr_amd import _lib
    from _hazard import recycle_scenario
    try:
        bad, rounds = recycle_scenario()               # (the scenario needs the spare slots: it switches them on)
    finally:
        _lib.set_global_option("keep_buffers", _session_mode())
    assert bad == 0, "a search in flight was disturbed in %d of %d rounds" % (bad, rounds)
