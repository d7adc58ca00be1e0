"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy/scipy) of the reference's retrieval hot path.

Nothing under oracle/ is product code.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it, and there only as the checker / the timed CPU baseline.
The product path (image-search-engine-for-historical-research_amd/) never imports it and
raises when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * a1 matching_L2, a2 IP ranker, a3 feature_enhancement/qge1, a7 l2n, a8 whitenapply,
    a9 compute_map2: PINNED against golden vectors produced by importing the reference's own
    functions in the build container (oracle/make_golden.py -> tests/golden/*.npz).
  * a4 diffusion branch, a5 Diffusion.get_offline_results, a6 faiss IndexFlatIP:
    PARITY UNPINNED -- faiss is absent and unpinned (requirements.txt:4 is commented out),
    the reference code needs scipy<1.12 / numpy<1.24.  The restatement follows
    src/utils/diffusion.py, src/utils/knn.py and src/utils/Reranking.py:230-253 as text with
    exact numpy inner-product top-k standing in for faiss.IndexFlatIP and scipy's cg(rtol=1e-6,
    atol=0) for the legacy cg(tol=1e-6).
"""
from .retrieval_oracle import *  # noqa: F401,F403
