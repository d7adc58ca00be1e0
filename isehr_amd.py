"""Import alias: the package directory `image-search-engine-for-historical-research_amd/` has a
name that is not a Python identifier, so `import isehr_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "image-search-engine-for-historical-research_amd")
_spec = importlib.util.spec_from_file_location(
    "isehr_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["isehr_amd"] = _mod
_spec.loader.exec_module(_mod)
