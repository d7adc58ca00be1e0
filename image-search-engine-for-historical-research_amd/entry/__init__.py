"""The build's own drivers with the reference's CLI surface for the retrieval path
(src/offline.py, src/online.py, src/test_rOP1m.py): same flags, one more --matching_method value
('HIP').  The CNN extractor is out of scope here: descriptors are read from the
reference's feature pickles (src/utils/general.py:67-92) or generated synthetically."""
