"""TEST INFRASTRUCTURE: runs the reference's own Python test suite (wrappers/python/tests) on the reference's own Python wrapper,
whose C extension (wrappers/python/src/zxc/_zxc.c) was compiled IN PLACE against this library's host sources over the mock device
(tests/c_abi/Makefile -> _bin/pywrap/_zxc*.so). Nothing of the reference is copied: the package's __init__.py is imported from
where it lies, the extension module is found next to this script's _bin/pywrap. Usage: run_ref_pytests.py <dir with _zxc*.so> [pytest args]"""
import importlib.util
import os
import sys

REF_PKG = "/root/reference/wrappers/python/src/zxc"
REF_TESTS = "/root/reference/wrappers/python/tests"


def main():
    ext_dir = os.path.abspath(sys.argv[1])
    spec = importlib.util.spec_from_file_location("zxc", os.path.join(REF_PKG, "__init__.py"), submodule_search_locations=[REF_PKG, ext_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["zxc"] = mod
    spec.loader.exec_module(mod)
    assert os.path.dirname(mod._zxc.__file__) == ext_dir, mod._zxc.__file__
    import pytest
    return pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir", ext_dir, REF_TESTS] + sys.argv[2:])


if __name__ == "__main__":
    sys.exit(main())
