"""Teacher-forced replay of a reference trajectory fixture on the HIP path, with the SCORE NETWORK'S OUTPUT of every step held
against what the reference recorded.

Why the network output has its own assertion: in the production configurations the predictor moves an atom by
g^2 s / sigma ~ 1e-6 of |X| per step (1e-8 at the bottom of the schedule) and a corrector by eps_i s / sigma < 1e-3 of |X| at
sigma = 0.2 (nothing at the bottom), so `torus_rel_l2(X) < 1e-5` after ONE step is passed by a network that returns zero scores
in every predictor step and holds the score to a few per cent at best in the correctors; and with one atom type
softmax([logit, -inf]) = [1, 0], so "A exact" does not depend on the logits either.  The reference records what its network returned in every step
(src/.../generators/langevin_generator.py:647-667,807-831: `model_predictions_i`), the fixtures hold it
(`pred_model_predictions_i_{A,X}`, `corr_model_predictions_i_{A,X}`), and `run` compares the HIP network's output with it step by
step.  `tests/test_egnn_c3_reference_gpu.py::test_teacher_forced_steps_fail_with_a_wrong_network` is the negative control.
"""
import numpy as np
import torch

from conftest import torus_rel_l2
from oracle import reference_sampler as RS


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def run(gen, spar, g, cuda, pinned=0):
    """Every predictor / corrector step of fixture `g` from the composition the reference recorded, with the reference's draws
    (gen.noise_source must replay them).  Returns one record per step:
        kind, k, index        "pred" | "corr", position in the fixture, time index
        a_equal               atom types of the step's output == the reference's
        x_err                 torus rel-L2 of the step's output coordinates
        pinned_equal          the first `pinned` rows of a predictor's output are the reference's bit for bit (repaint)
        score_err             rel-L2 of the network's X output against the reference's recorded one
        score_exact, floor    (when the fixture holds the reference module's binary64 output) distance of the HIP output from
                              it, and the reference's own binary32 output's distance from it
        logits_close          finite logits within rtol 1e-4 / atol 1e-5 of the recorded ones, MASK logit -inf
    """
    B, M = int(g["batch"]), spar.number_of_corrector_steps
    seen = []
    inner = gen._get_model_predictions

    def spy(*args, **kwargs):
        out = inner(*args, **kwargs)
        seen.append((out.A.detach().cpu().numpy().copy(), out.X.detach().cpu().numpy().copy()))
        return out

    gen._get_model_predictions = spy

    def axl(prefix, k):
        return RS.AXL(A=torch.from_numpy(g[prefix + "_A"][k]).to(cuda), X=torch.from_numpy(g[prefix + "_X"][k]).to(cuda),
                      L=torch.from_numpy(g[prefix + "_L"][k]).to(cuda))

    def record(kind, k, index, out, want_a, want_x):
        assert len(seen) == 1, "one network forward per step"
        logits, scores = seen.pop()
        ref_logits, ref_scores = g[f"{kind}_model_predictions_i_A"][k], g[f"{kind}_model_predictions_i_X"][k]
        rec = dict(kind=kind, k=k, index=int(index), a_equal=np.array_equal(out.A.cpu().numpy(), want_a),
                   x_err=torus_rel_l2(out.X.cpu().numpy(), want_x), score_err=_rel(scores, ref_scores),
                   logits_close=bool(np.isneginf(logits[..., -1]).all() and np.allclose(
                       logits[..., :-1], ref_logits[..., :-1], rtol=1e-4, atol=1e-5)))
        key64 = f"{kind}_model_predictions_i_X_fp64"
        if key64 in g.files:
            rec["score_exact"], rec["floor"] = _rel(scores, g[key64][k]), _rel(ref_scores, g[key64][k])
        if pinned and kind == "pred":
            got = out.X.cpu().numpy()
            rec["pinned_equal"] = np.array_equal(got[:, :pinned].view(np.int32), want_x[:, :pinned].view(np.int32))
        return rec

    records = []
    try:
        with torch.no_grad():
            gen._prepare(cuda)
            gen._begin_call(cuda)
            forces = torch.zeros(B, spar.number_of_atoms, 3, device=cuda)
            for k, index in enumerate(g["pred_index"]):
                out = gen.predictor_step(axl("pred_composition_i", k), int(index), forces)
                records.append(record("pred", k, index, out, g["pred_composition_im1_A"][k], g["pred_composition_im1_X"][k]))
                for m in range(M):
                    kk = k * M + m
                    out = gen.corrector_step(axl("corr_composition_i", kk), int(index) - 1, forces, m)
                    records.append(record("corr", kk, int(index) - 1, out, g["corr_corrected_composition_i_A"][kk],
                                          g["corr_corrected_composition_i_X"][kk]))
    finally:
        gen._get_model_predictions = inner
    gen.check_status()
    assert gen.noise_source.inner.exhausted()
    return records


def check(records, label, floor_rule=False):
    """The assertions of a teacher-forced test.  `floor_rule` (N = 216 in the 16.5 A graph cell, where the reference's own
    binary32 output is 2.3e-5 from its binary64 evaluation): the score bar is max(1e-5, floor) AND the HIP output must be no
    further from the binary64 evaluation than 1.05 x the reference itself is."""
    for r in records:
        where = f"{label} {r['kind']} step {r['k']} (time index {r['index']})"
        assert r["a_equal"], f"{where}: atom types differ from the reference"
        assert r.get("pinned_equal", True), f"{where}: repainted rows differ in bits"
        assert r["x_err"] < 1e-5, f"{where}: coordinates rel-L2 {r['x_err']:.2e}"
        bar = max(1e-5, r["floor"]) if floor_rule else 1e-5
        assert r["score_err"] < bar, f"{where}: network output (scores) rel-L2 {r['score_err']:.2e} against the reference's"
        if floor_rule:
            assert r["score_exact"] < 1.05 * r["floor"], \
                f"{where}: network output {r['score_exact']:.2e} from the exact evaluation (the reference itself: {r['floor']:.2e})"
        assert r["logits_close"], f"{where}: network output (logits) differs from the reference's"


def summary(records):
    worst = max(records, key=lambda r: r["score_err"])
    return (f"{len(records)} steps: worst X {max(r['x_err'] for r in records):.2e}, worst score {worst['score_err']:.2e} "
            f"({worst['kind']} {worst['k']}, index {worst['index']})")


def forward_check(scores, g, label):
    """Assertions of a forward fixture that holds the reference's binary32 output (`out_X`) and its module's binary64 output on
    the same inputs (`out_X_fp64`):
      * over the batch: rel-L2 <= 1e-5 against the reference;
      * per structure: rel-L2 <= max(2e-5, 1.5 floor_b), floor_b = the reference's own distance from its binary64 evaluation on
        that structure.  Why a floor: near a symmetric configuration (the diamond sites displaced by sigma = 1e-4: the end of a
        trajectory) the score is a difference that nearly cancels -- |score| ~ 3e-6 against 2e-3 for a random structure -- while
        the rounding of every graph layer's x + trans stays ~3e-8 absolute: the REFERENCE's binary32 output is 1 % from its own
        binary64 evaluation there (and 1e-3 at sigma = 1e-3, 1e-4 at 1e-2), so no relative bar below that means anything.
        Why 1.5: two binary32 evaluations whose rounding is independent and f from the exact answer sit sqrt(2) f apart; the
        measured ratios on these structures are 1.0 - 1.2 (part of the rounding is shared).
    Returns (batch error, worst structure error, its index, batch floor) for the caller's report."""
    got, ref, ref64 = (np.asarray(a, np.float64) for a in (scores, g["out_X"], g["out_X_fp64"]))
    rows = lambda a: np.linalg.norm(a.reshape(len(a), -1), axis=1)         # noqa: E731
    err, per, floor = _rel(got, ref), rows(got - ref) / rows(ref), rows(ref - ref64) / rows(ref64)
    assert err < 1e-5, f"{label}: scores rel-L2 {err:.2e} against the reference"
    bad = per > np.maximum(2e-5, 1.5 * floor)
    assert not bad.any(), f"{label}: structures {np.nonzero(bad)[0].tolist()}: {per[bad]} (the reference's own floor {floor[bad]})"
    return err, float(per.max()), int(per.argmax()), _rel(ref, ref64)
