"""Forward records crossing between the fp64 oracle and the engine — test helpers of the "same record on both sides" parity tests.

The reference's stepBackward differentiates the ForwardInformation it is handed (Simulation.cpp:1455-1551). Two statements isolate the
adjoint kernels from the forward iterate (two fp PD loops never stop at bitwise the same state, and on near-singular adjoint systems
that difference alone moves the gradient by more than 1e-4):
  * teacher forcing: the ORACLE's record is uploaded with dc_set_record and the engine differentiates it (`upload_oracle_records`);
  * adoption: the oracle takes over the ENGINE's record of the step — x_new as the adjoint kernel re-forms it, f, the contact
    normals — and differentiates that (`oracle_adopts_gpu_record`).
Both compare gradients of the SAME linear system; the gate is BASELINE.json's flat 1e-4.
"""
import numpy as np


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-30)


def oracle_record(o, ref, prim_map=None):
    """Record `ref` (the dict Oracle.step returned) as the per-rollout arrays dc_set_record takes. prim_map: oracle primitive id ->
    index into the engine's primitive list (identity by default; a LowerLeg collection has no single index)."""
    N = o.N
    f, r = o.record_fr(ref["id"])
    pc = o.prim_contacts(ref["id"])
    prim = -np.ones(N, dtype=np.int32)
    normal = np.zeros((N, 3))
    for k in range(len(pc["particle"])):
        if pc["prim"][k] < 0:
            continue
        prim[pc["particle"][k]] = pc["prim"][k] if prim_map is None else prim_map[pc["prim"][k]]
        normal[pc["particle"][k]] = pc["normal"][k]
    sc = o.self_contacts(ref["id"])
    selfc = dict(pairs=np.stack([sc["p1"], sc["p2"]], axis=1) if len(sc["p1"]) else np.zeros((0, 2), dtype=np.int32),
                 layer=sc["layer"], normal=sc["normal"], d=sc["d"])
    return dict(x=ref["x"], v=ref["v"], f=f, r=r, prim=prim, normal=normal.reshape(-1), self=selfc)


def upload_oracle_records(e, slot, recs, x_fixed=None):
    """dc_set_record for a batch: recs = one oracle_record per rollout of the engine's batch."""
    e.set_record(slot, np.stack([q["x"] for q in recs]), np.stack([q["v"] for q in recs]), np.stack([q["f"] for q in recs]),
                 np.stack([q["prim"] for q in recs]), np.stack([q["normal"] for q in recs]), r=np.stack([q["r"] for q in recs]),
                 x_fixed=x_fixed, self_contacts=[q["self"] for q in recs])


def gpu_xnew64(x_prev, x_new, v_new, h):
    """x_new as the adjoint kernel takes it (xnew64, dc_adjoint64.h): the unrounded x_prev + h v_new re-formed in fp64 from its fp32
    terms wherever the tape's value is the fp32 rounding of that sum, the stored value elsewhere (a step that reverted to its best iterate)."""
    xs = np.asarray(x_new, dtype=np.float64)
    xh = np.asarray(x_prev, dtype=np.float64) + np.asarray(v_new, dtype=np.float64) * float(h)
    ok = np.abs(xh - xs) <= 2.4e-7 * np.maximum(np.abs(xs), 1e-3)
    return np.where(ok, xh, xs)


def oracle_adopts_gpu_record(o, rid, e, slot, b, x_prev, x_new, v_new, f, h, normals=None, self_cap=16384):
    """The oracle's record `rid` becomes the engine's record of rollout b (the contact SETS must already agree): x_new, f, primitive-contact
    normals (normals = e.get_contacts(slot)[1], fetched once by the caller) and self-contact normals."""
    if normals is None:
        normals = e.get_contacts(slot)[1]
    sc = e.get_self_contacts(slot, rollout=b, cap=self_cap)
    pairs = sc["pairs"] if sc["count"] else None
    return o.adopt_record(rid, gpu_xnew64(x_prev, x_new, v_new, h), f, prim_normal=normals[b], self_pairs=pairs,
                          self_normal=sc["normal"] if sc["count"] else None)


def mu_err(g, r):
    """relative error of dL/dmu (a scalar per friction group); 0 where the reference value is negligible"""
    g, r = np.asarray(g, dtype=np.float64).reshape(-1), np.asarray(r, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(g - r) / np.maximum(np.abs(r), 1e-30) * (np.abs(r) > 1e-9)))


def stopping_sensitivity(o, x0, v0, xf, iters, gx, gv, rb, quantity=lambda r: r["dL_dmu"]):
    """How much the oracle's OWN gradient output moves when its PD loop is stopped one iteration later than its stopping rule did
    (tolerance off, cap = iters + 1, diagnostic 4 = the capped loop keeps its last iterate): the forward record is a PD iterate, and an
    output that moves by more than the gate under one more iteration is not defined to the gate by the step's inputs — the end-to-end
    comparison of two fp PD loops can then only be a conditioning report. Restores the oracle's settings."""
    saved_tol, saved_cap = o.params["fwd_tol"], o.flags["cap"]
    try:
        o.set(fwd_tol=0.0, cap=int(iters) + 1); o.build(); o.diagnostics(4)
        ref2 = o.step(x0, v0, xf)
        rb2 = o.step_backward(ref2["id"], gx, gv, is_start=False, direct=True)
    finally:
        o.set(fwd_tol=saved_tol, cap=saved_cap); o.build(); o.diagnostics(0)
    return mu_err(quantity(rb2), quantity(rb))
