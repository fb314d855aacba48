"""The PUBLIC surface of the reference's modules against this package's modules of the same relative path (container-only:
imports /root/reference through make_golden's stubs).

    PYTHONPATH=/root/reference/src python tests/golden/api_surface.py > report.json

For every module of this package that has a counterpart at the same relative path in the reference: every public (no leading
underscore) class, function and method the reference module DEFINES must exist here; dataclass fields must exist with the same
defaults; the parameters of public callables must carry the reference's names, order and defaults (this package may append
parameters of its own).  Private helpers (leading underscore) are this package's own business -- except the ones listed in
PRIVATE_BUT_SERVED, which the reference's tests and subclasses call.  Prints one JSON report; tests/test_reference_yaml_surface.py
holds the (short, reasoned) list of differences that are accepted."""
import dataclasses
import importlib
import inspect
import json
import os
import pkgutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden  # noqa: E402,F401  (installs the container-only stubs, imports the reference)

OWN, REF = "diffusion_for_multi_scale_molecular_dynamics_amd", "diffusion_for_multi_scale_molecular_dynamics"
PRIVATE_BUT_SERVED = {"_get_model_predictions", "_relative_coordinates_update", "_atom_types_update", "_lattice_parameters_update",
                      "_draw_coordinates_gaussian_sample", "_draw_lattice_gaussian_sample", "_draw_gumbel_sample", "_draw_binary_sample",
                      "_forward_unchecked", "_check_batch"}


def parameters(fn):
    try:
        return [(p.name, None if p.default is inspect.Parameter.empty else repr(p.default))
                for p in inspect.signature(fn).parameters.values()]
    except (TypeError, ValueError):
        return None


def compare_callable(where, ref_fn, own_fn, report):
    ref_p, own_p = parameters(ref_fn), parameters(own_fn)
    if ref_p is None or own_p is None:
        return
    names_match = [n for n, _ in ref_p] == [n for n, _ in own_p][:len(ref_p)]
    # (a parameter the reference REQUIRES may be optional here: an extension, not a difference)
    defaults_match = all(default is None or default == dict(own_p).get(name) for name, default in ref_p)
    if not (names_match and defaults_match):
        report["signature_differences"].append(dict(where=where, reference=ref_p, own=own_p))


def syntax_parameters(node):
    import ast
    args = node.args
    positional = args.posonlyargs + args.args
    defaults = [None] * (len(positional) - len(args.defaults)) + [ast.unparse(d) for d in args.defaults]
    out = list(zip([a.arg for a in positional], defaults))
    out += [(a.arg, None if d is None else ast.unparse(d)) for a, d in zip(args.kwonlyargs, args.kw_defaults)]
    return out


def compare_by_syntax(rel, reference_source, own_module, report):
    """The same comparison from the reference module's source text (it cannot be imported here): public functions, classes and
    their public methods, parameter names and order; defaults compared as source text after `ast.unparse` on both sides."""
    import ast
    tree = ast.parse(open(reference_source).read())
    own_tree = ast.parse(open(own_module.__file__).read())
    own_functions = {n.name: n for n in own_tree.body if isinstance(n, ast.FunctionDef)}
    own_classes = {n.name: n for n in own_tree.body if isinstance(n, ast.ClassDef)}

    def compare(where, ref_node, own_node):
        ref_p, own_p = syntax_parameters(ref_node), syntax_parameters(own_node)
        names_match = [n for n, _ in ref_p] == [n for n, _ in own_p][:len(ref_p)]
        # (a parameter the reference REQUIRES may be optional here: an extension, not a difference)
        defaults_match = all(default is None or default == dict(own_p).get(name) for name, default in ref_p)
        if not (names_match and defaults_match):
            report["signature_differences"].append(dict(where=where, reference=ref_p, own=own_p))

    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and not node.name.startswith("_"):
            if node.name not in own_functions and not hasattr(own_module, node.name):
                report["public_missing"].append(f"{rel}.{node.name}")
            elif node.name in own_functions:
                compare(f"{rel}.{node.name}", node, own_functions[node.name])
        elif isinstance(node, ast.ClassDef) and not node.name.startswith("_"):
            if node.name not in own_classes and not hasattr(own_module, node.name):
                report["public_missing"].append(f"{rel}.{node.name}")
                continue
            own_class = own_classes.get(node.name)
            own_object = getattr(own_module, node.name)
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and (not item.name.startswith("_") or item.name in PRIVATE_BUT_SERVED | {"__init__"}):
                    if not hasattr(own_object, item.name):
                        report["public_missing"].append(f"{rel}.{node.name}.{item.name}")
                    elif own_class is not None:
                        mine = [m for m in own_class.body if isinstance(m, ast.FunctionDef) and m.name == item.name]
                        if mine:
                            compare(f"{rel}.{node.name}.{item.name}", item, mine[0])
                elif isinstance(item, ast.AnnAssign) and isinstance(item.target, ast.Name) and not item.target.id.startswith("_"):
                    # a dataclass field: it must exist here with the same default
                    import dataclasses as dc
                    if dc.is_dataclass(own_object):
                        fields = {f.name: f for f in dc.fields(own_object)}
                        if item.target.id not in fields:
                            report["dataclass_differences"].append(dict(where=f"{rel}.{node.name}", missing=[item.target.id]))
                        elif item.value is not None and not isinstance(item.value, ast.Call):
                            want = ast.literal_eval(item.value) if isinstance(item.value, (ast.Constant, ast.List, ast.Tuple, ast.UnaryOp)) else None
                            if want is not None and fields[item.target.id].default != want:
                                report["dataclass_differences"].append(dict(where=f"{rel}.{node.name}", default_differs=[item.target.id]))


def main():
    import diffusion_for_multi_scale_molecular_dynamics_amd as own_package
    report = dict(modules_compared=[], modules_compared_by_syntax=[], modules_without_counterpart=[], public_missing=[], served_private_missing=[],
                  signature_differences=[], dataclass_differences=[])
    for info in sorted(pkgutil.walk_packages(own_package.__path__, OWN + "."), key=lambda m: m.name):
        rel = info.name[len(OWN):]
        own_module = importlib.import_module(info.name)
        if info.ispkg:
            continue
        try:
            ref_module = importlib.import_module(REF + rel)
        except ImportError as exc:
            # absent in the reference (this package's own modules), or not importable here (it needs torchode / mace / lightning /
            # pymatgen / orion): then its SOURCE TEXT is parsed instead (names, parameters, defaults)
            source = os.path.join(os.path.dirname(importlib.import_module(REF).__file__), *rel.strip(".").split(".")) + ".py"
            if os.path.isfile(source):
                compare_by_syntax(rel, source, own_module, report)
                report["modules_compared_by_syntax"].append(rel)
            else:
                report["modules_without_counterpart"].append(dict(module=rel, why=f"{type(exc).__name__}: {exc}"[:120]))
            continue
        report["modules_compared"].append(rel)
        for name, ref_obj in vars(ref_module).items():
            if getattr(ref_obj, "__module__", None) != REF + rel or not (inspect.isclass(ref_obj) or inspect.isfunction(ref_obj)):
                continue
            if name.startswith("_"):
                continue
            own_obj = getattr(own_module, name, None)
            if own_obj is None:
                report["public_missing"].append(f"{rel}.{name}")
                continue
            if inspect.isfunction(ref_obj):
                compare_callable(f"{rel}.{name}", ref_obj, own_obj, report)
                continue
            if dataclasses.is_dataclass(ref_obj):
                if not dataclasses.is_dataclass(own_obj):
                    report["dataclass_differences"].append(dict(where=f"{rel}.{name}", problem="not a dataclass here"))
                else:
                    ref_f = {f.name: repr(f.default) for f in dataclasses.fields(ref_obj)}
                    own_f = {f.name: repr(f.default) for f in dataclasses.fields(own_obj)}
                    missing = [k for k in ref_f if k not in own_f]
                    differing = [k for k in ref_f if k in own_f and ref_f[k] != own_f[k]]
                    if missing or differing:
                        report["dataclass_differences"].append(dict(where=f"{rel}.{name}", missing=missing, default_differs=differing))
            for method, ref_m in vars(ref_obj).items():
                is_dunder = method.startswith("__")
                if is_dunder and method != "__init__":
                    continue
                if not (callable(ref_m) or isinstance(ref_m, (staticmethod, classmethod, property))):
                    continue
                if method.startswith("_") and not is_dunder and method not in PRIVATE_BUT_SERVED:
                    continue
                if not hasattr(own_obj, method):
                    key = "served_private_missing" if method.startswith("_") else "public_missing"
                    report[key].append(f"{rel}.{name}.{method}")
                    continue
                if isinstance(ref_m, property):
                    continue
                if method == "__init__" and dataclasses.is_dataclass(ref_obj):
                    continue
                compare_callable(f"{rel}.{name}.{method}", getattr(ref_obj, method), getattr(own_obj, method), report)
    report["helper_values"] = helper_values()
    print(json.dumps(report))


def helper_values():
    """The small helpers that now live under the reference's module paths, evaluated on both sides (True = equal)."""
    import torch

    def both(rel, name):
        return getattr(importlib.import_module(REF + rel), name), getattr(importlib.import_module(OWN + rel), name)

    out = {}
    ref_f, own_f = both(".utils.lattice_utils", "get_relative_coordinates_lattice_vectors")
    out["relative_coordinates_lattice_vectors"] = all(torch.equal(ref_f(n, d), own_f(n, d)) and ref_f(n, d).dtype == own_f(n, d).dtype
                                                      for n in (1, 2) for d in (1, 2, 3))
    ref_f, own_f = both(".utils.lattice_utils", "get_cubic_point_group_complete_lattice_shells")
    out["complete_lattice_shells"] = all(
        len(ref_f(n, d)) == len(own_f(n, d)) and all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(ref_f(n, d), own_f(n, d)))
        for n in (1, 2, 3) for d in (1, 2, 3))
    ref_f, own_f = both(".utils.lattice_utils", "get_cubic_point_group_positive_normalized_bloch_wave_vectors")
    out["positive_bloch_wave_vectors"] = all(torch.equal(ref_f(n, d), own_f(n, d)) and ref_f(n, d).dtype == own_f(n, d).dtype for n in (1, 2, 3) for d in (1, 2, 3))
    ref_f, own_f = both(".utils.geometric_utils", "get_cubic_point_group_symmetries")
    out["cubic_point_group_symmetries"] = all(torch.equal(ref_f(d), own_f(d)) and ref_f(d).dtype == own_f(d).dtype for d in (1, 2, 3))
    ref_f, own_f = both(".utils.symmetry_utils", "get_all_permutation_indices")
    out["permutation_indices"] = all(all(torch.equal(a, b) and a.dtype == b.dtype for a, b in zip(ref_f(n), own_f(n))) for n in (1, 2, 4))
    g0 = torch.Generator().manual_seed(5)
    values, matrices = torch.rand(3, generator=g0), torch.rand(3, 2, 4, generator=g0)
    ref_f, own_f = both(".utils.tensor_utils", "broadcast_batch_tensor_to_all_dimensions")
    ref_m, own_m = both(".utils.tensor_utils", "broadcast_batch_matrix_tensor_to_all_dimensions")
    out["tensor_utils"] = all(torch.equal(ref_f(values, shape), own_f(values, shape)) and torch.equal(ref_m(matrices, shape), own_m(matrices, shape))
                              for shape in ((3,), (3, 5), (3, 2, 7)))
    times = torch.linspace(0.0, 1.0, 33)
    ref_f, own_f = both(".noise_schedulers.sigma_calculator", "instantiate_sigma_calculator")
    out["sigma_calculators"] = all(
        torch.equal(ref_f(lo, hi, kind)(times), own_f(lo, hi, kind)(times)) and
        torch.equal(ref_f(lo, hi, kind).get_sigma_time_derivative(times), own_f(lo, hi, kind).get_sigma_time_derivative(times)) and
        list(ref_f(lo, hi, kind).state_dict()) == list(own_f(lo, hi, kind).state_dict())
        for kind in ("exponential", "linear") for lo, hi in ((1e-4, 0.25), (1e-3, 0.5), (1e-4, 0.2)))
    g = torch.Generator().manual_seed(8)
    data, ids = torch.randn(40, 5, generator=g), torch.randint(0, 9, (40,), generator=g)
    for name in ("unsorted_segment_sum", "unsorted_segment_mean"):
        ref_f, own_f = both(".models.egnn_utils", name)
        out[name] = bool(torch.allclose(ref_f(data, ids, 11), own_f(data, ids, 11), rtol=1e-6, atol=1e-6))
    ref_f, own_f = both(".models.egnn_utils", "get_edges")
    out["get_edges"] = all(ref_f(n) == own_f(n) for n in (1, 2, 5))
    ref_f, own_f = both(".models.egnn_utils", "get_edges_batch")
    out["get_edges_batch"] = all(torch.equal(ref_f(n, b), own_f(n, b)) for n, b in ((2, 1), (4, 3)))
    ref_c, own_c = both(".data.element_types", "ElementTypes")
    elements = ["Si", "Ge", "C"]
    a, b = ref_c(elements), own_c(elements)
    out["element_types"] = (a.elements == b.elements and a.element_ids == b.element_ids and a.number_of_atom_types == b.number_of_atom_types and
                            all(a.get_element(k) == b.get_element(k) for k in (-1, 0, 1, 2)) and
                            all(a.get_element_id(e) == b.get_element_id(e) for e in elements + [a.get_element(-1)]))
    cell = torch.diag_embed(torch.rand(4, 3, generator=g) + 4.0) + 0.1 * torch.rand(4, 3, 3, generator=g)
    x = torch.rand(4, 6, 3, generator=g)
    checks = []
    for name, args in (("get_reciprocal_basis_vectors", (cell,)), ("get_positions_from_coordinates", (x, cell)),
                       ("get_relative_coordinates_from_cartesian_positions", (x, cell)),
                       ("map_unit_cell_to_lattice_parameters", (torch.diag_embed(torch.rand(4, 3, generator=g)),)),
                       ("map_lattice_parameters_to_unit_cell_vectors", (torch.tensor([[4.0, 5.0, 6.0, 0.0, 0.0, 0.0]]),)),
                       ("map_noisy_axl_lattice_parameters_to_unit_cell_vectors", (torch.tensor([[2.0, 5.0, 7.0, 0.3, -0.2, 9.0]]),))):
        ref_f, own_f = both(".utils.basis_transformations", name)
        checks.append(bool(torch.equal(ref_f(*[a.clone() for a in args]), own_f(*[a.clone() for a in args]))))
    ref_f, own_f = both(".utils.basis_transformations", "get_spatial_dimension_from_number_of_lattice_parameters")
    checks.append(all(ref_f(k) == own_f(k) for k in (1, 3, 6)))
    ref_f, own_f = both(".utils.basis_transformations", "map_numpy_unit_cell_to_lattice_parameters")
    import numpy as np
    checks.append(bool(np.array_equal(ref_f(np.diag([1.0, 2.0, 3.0])), own_f(np.diag([1.0, 2.0, 3.0])))))
    out["basis_transformations"] = all(checks)
    return out


if __name__ == "__main__":
    main()
