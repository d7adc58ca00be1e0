"""MI355X-native retrieval hot path (normalise/whiten -> exhaustive kNN top-K -> alpha-QE /
diffusion re-rank) behind the reference's `matching_<method>` / `qge1` / `QGE` surface.

Import as `isehr_amd` (see /isehr_amd.py).  Heavy sub-modules load the HIP C-ABI library
(libmi355_retrieval.so) on first use and raise if it is missing: there is no CPU fallback.
"""
__version__ = "0.1.0"
