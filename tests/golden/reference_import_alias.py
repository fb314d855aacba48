"""pytest plugin (container-only): `import diffusion_for_multi_scale_molecular_dynamics.<path>` resolves to THIS package's module
of the same relative path, so that the reference's own test files can be run against this package where they need no GPU
(tests/test_reference_yaml_surface.py::test_the_references_own_tests_of_the_host_side_helpers_pass_here):
    PYTHONPATH=tests/golden:. python -m pytest -p reference_import_alias /root/reference/tests/utils/test_lattice_utils.py"""
import importlib, importlib.abc, importlib.util, sys
REF, OWN = "diffusion_for_multi_scale_molecular_dynamics", "diffusion_for_multi_scale_molecular_dynamics_amd"
class Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name == REF or name.startswith(REF + "."):
            return importlib.util.spec_from_loader(name, self)
        return None
    def create_module(self, spec):
        module = importlib.import_module(OWN + spec.name[len(REF):])
        return module
    def exec_module(self, module):
        pass
sys.meta_path.insert(0, Alias())
