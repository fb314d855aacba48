"""pytest plugin (container-only): `import diffusion_for_multi_scale_molecular_dynamics.<path>` -- and the `src.`-prefixed form some of
the reference's test files use -- resolves to THIS package's module of the same relative path, so that the reference's own test
files can be run against this package where they need no GPU
(tests/test_reference_yaml_surface.py::test_the_references_own_tests_of_the_host_side_helpers_pass_here):
    PYTHONPATH=tests/golden:. python -m pytest -p reference_import_alias /root/reference/tests/utils/test_lattice_utils.py
At the end of the session the plugin checks that NO module of the reference's source tree was imported (exit status 3 if one was:
the run would have tested the reference, not this package)."""
import importlib
import importlib.abc
import importlib.util
import sys
import types

REF, OWN = "diffusion_for_multi_scale_molecular_dynamics", "diffusion_for_multi_scale_molecular_dynamics_amd"
PREFIXES = (REF, "src." + REF)


class Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name == "src" or any(name == p or name.startswith(p + ".") for p in PREFIXES):
            return importlib.util.spec_from_loader(name, self, is_package=True)
        return None

    def create_module(self, spec):
        if spec.name == "src":
            module = types.ModuleType("src")
            module.__path__ = []
            return module
        prefix = next(p for p in PREFIXES if spec.name == p or spec.name.startswith(p + "."))
        return importlib.import_module(OWN + spec.name[len(prefix):])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, Alias())


def pytest_sessionfinish(session, exitstatus):
    leaked = sorted(name for name, module in list(sys.modules.items())
                    if (getattr(module, "__file__", None) or "").startswith("/root/reference/src"))
    if leaked:
        print("\nREFERENCE MODULES WERE IMPORTED:", leaked[:5])
        session.exitstatus = 3
