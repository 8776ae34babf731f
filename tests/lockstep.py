"""Lockstep replay of a GPU solve against the reference's L-BFGS (test logic, shared with bench.py's checker leg).

Input: the evaluation trace of one trajectory (dftpav_batch_trace: every x the kernel evaluated with its f, g, the
direction d in force, the trial step, the iteration number) and a LITERAL evaluator — the reference build oracle/_ref, or
the literal oracle that is bit-equal to it.  The replay walks lbfgs_optimize / line_search_lewisoverton (lbfgs.hpp:276-390,
524-745) along the GPU's own evaluation points and checks, statement by statement:
  * every evaluated point: literal f, g against the kernel's                                   (rel_f, rel_g)
  * every trial point is xp + stp d, every trial step is the one lbfgs.hpp's update rule produces
  * every branch of the line search (early exit :326-329, Armijo :331-335, weak Wolfe :339-346) and of the outer loop
    (convergence :628-635, past/delta stop :642-659, cautious update :704-706), decided from the LITERAL values, is the
    branch the kernel took — until the first branch whose margin is within rounding (`flip`), where the replay ends
  * every search direction: the plain sequential two-loop recursion (lbfgs.hpp:716-739) over the kernel's own history
    against the kernel's blocked, fused recursion                                              (rel_d)
"""
import numpy as np


def two_loop(S, Y, YS, g, order):
    """lbfgs.hpp:716-739 with sequential dots; `order` lists the stored pairs newest first"""
    d = -g.copy()
    alpha = {}
    for j in order:
        a = float(np.dot(S[j], d)) / YS[j]
        alpha[j] = a
        d += (-a) * Y[j]
    newest = order[0]
    d *= YS[newest] / float(np.dot(Y[newest], Y[newest]))
    for j in reversed(order):
        beta = float(np.dot(Y[j], d)) / YS[j]
        d += (alpha[j] - beta) * S[j]
    return d


def replay(tr, lit_eval, p, flip_tol=1e-9, direction_every=1):
    """tr: dict from Batch.get_trace(); lit_eval(x) -> (f, g); p: dftpav_params.  Returns a report dict; raises
    AssertionError on a branch mismatch whose margin exceeds flip_tol.  direction_every: the plain two-loop recursion is
    replayed (in Python, the slow part) at every such iteration only; every other check is made at every evaluation."""
    X, G, D, F, STP, K = tr["x"], tr["g"], tr["d"], tr["f"], tr["stp"], tr["k"]
    E = len(F)
    fl = np.zeros(E)
    gl = np.zeros_like(G)
    for e in range(E):
        fl[e], gl[e] = lit_eval(X[e])
    rel_f = float(np.max(np.abs(F - fl) / np.maximum(1.0, np.abs(fl))))
    rel_g = float(np.max(np.max(np.abs(G - gl), axis=1) / np.maximum(1.0, np.max(np.abs(gl), axis=1))))
    rep = dict(evals=E, rel_f=rel_f, rel_g=rel_g, rel_d=0.0, rel_x=0.0, branches=0, iterations=0, flip=None, status=None,
               min_margin=np.inf)
    m, past = p.lbfgs_mem_size, p.lbfgs_past

    def branch(name, lit, gpu, margin, e):
        rep["branches"] += 1
        rep["min_margin"] = min(rep["min_margin"], margin)
        if lit != gpu:
            assert margin <= flip_tol, "branch %s differs at evaluation %d with margin %.3e (literal %s, kernel %s)" % (name, e, margin, lit, gpu)
            rep["flip"] = dict(eval=e, branch=name, margin=margin)
            return False
        return True

    # lbfgs.hpp:524-551
    xp, gp_gpu = X[0].copy(), G[0].copy()
    fx_lit = fl[0]
    pf = [fl[0]] + [0.0] * max(0, past - 1)
    if E == 1:
        rep["status"] = 0
        return rep
    step = 1.0 / np.sqrt(float(np.dot(G[0], G[0])))
    d_ref = -G[0]
    S, Y, YS, order = {}, {}, {}, []
    end = 0
    e = 1
    k = 1
    while e < E:
        rows = [r for r in range(e, E) if K[r] == k]  # evaluations of this iteration's line search
        assert rows and rows[0] == e, "trace rows out of order"
        d = D[e]
        if d_ref is not None:
            rep["rel_d"] = max(rep["rel_d"], float(np.max(np.abs(d - d_ref)) / max(np.max(np.abs(d_ref)), 1e-300)))
        gp_lit = lit_eval(xp)[1]
        finit = fx_lit
        dginit = float(np.dot(gp_lit, d))
        dgtest, dstest = p.lbfgs_f_dec_coeff * dginit, p.lbfgs_s_curv_coeff * dginit
        mu, nu, brackt = 0.0, p.lbfgs_max_step, False
        stp = step
        accepted = None
        for r in rows:
            assert abs(STP[r] - stp) <= 1e-9 * abs(stp), "trial step %r, expected %r at evaluation %d" % (STP[r], stp, r)
            stp = STP[r]
            rep["rel_x"] = max(rep["rel_x"], float(np.max(np.abs(X[r] - (xp + stp * d)) / np.maximum(1.0, np.abs(X[r])))))
            last = r == rows[-1]
            more = r + 1 < E
            # what the kernel did with this trial
            if not last:
                nxt = STP[r + 1]
                arm = 0.5 * (mu + stp)
                wol = 0.5 * (stp + nu) if brackt else 2.0 * stp
                if abs(nxt - arm) <= 1e-9 * abs(arm) and abs(nxt - wol) > 1e-9 * abs(wol):
                    gpu = "armijo"
                elif abs(nxt - wol) <= 1e-9 * abs(wol):
                    gpu = "wolfe"
                else:
                    gpu = "clamped"
            else:
                gpu = "accept" if more else "end"
            # what the reference does with the literal values
            f = fl[r]
            early_lhs = abs(finit - f) / (abs(finit) + 1.0)
            early_rhs = p.lbfgs_delta / past if past > 0 else -1.0
            m0 = abs(early_lhs - early_rhs) / max(early_rhs, 1e-300)
            if past > 0 and early_lhs < early_rhs:
                lit, margin = "accept", m0
            elif f > finit + stp * dgtest:
                lit, margin = "armijo", min(abs(f - (finit + stp * dgtest)) / max(1.0, abs(f)), m0)
            else:
                gs = float(np.dot(gl[r], d))
                m1 = abs(f - (finit + stp * dgtest)) / max(1.0, abs(f))
                m2 = abs(gs - dstest) / max(abs(dstest), 1e-300)
                lit, margin = ("wolfe" if gs < dstest else "accept"), min(m0, m1, m2)
            if gpu == "end":
                # the last recorded evaluation: the solve ended here (accepted and stopped, or a failed search)
                rep["status"] = "ended after %s" % lit
                if lit != "accept":
                    rep["iterations"] = k
                    return rep
                accepted = r
                break
            if gpu == "clamped":
                return rep  # step bounds touched (lbfgs.hpp:370-388): not replayed further
            if not branch("line search", lit, gpu, margin, r):
                return rep
            if lit == "accept":
                accepted = r
                break
            if lit == "armijo":
                nu, brackt = stp, True
            else:
                mu = stp
            stp = 0.5 * (mu + nu) if brackt else 2.0 * stp
        if accepted is None:
            return rep
        r = accepted
        rep["iterations"] = k
        fx_lit = fl[r]
        # lbfgs.hpp:628-666 on the literal values
        cont_gpu = r + 1 < E
        gn, xn = float(np.max(np.abs(gl[r]))), float(np.max(np.abs(X[r])))
        if gn / max(1.0, xn) < p.lbfgs_g_epsilon:
            branch("convergence", "stop", "go" if cont_gpu else "stop", 1.0, r)
            rep["status"] = 0
            return rep
        stop = False
        if past > 0:
            if past <= k:
                rate = abs(pf[k % past] - fx_lit) / max(1.0, abs(fx_lit))
                stop = rate < p.lbfgs_delta
                if not branch("past/delta stop", "stop" if stop else "go", "go" if cont_gpu else "stop", abs(rate - p.lbfgs_delta) / p.lbfgs_delta, r):
                    return rep
            pf[k % past] = fx_lit
        if stop or not cont_gpu:
            rep["status"] = 1 if stop else rep["status"]
            return rep
        k += 1
        # history update from the kernel's own iterates (lbfgs.hpp:676-706), then the reference's plain recursion
        s, y = X[r] - xp, G[r] - gp_gpu
        ys = float(np.dot(y, s))
        cau = float(np.dot(s, s)) * np.sqrt(float(np.dot(gp_gpu, gp_gpu))) * p.lbfgs_cautious_factor
        S[end], Y[end], YS[end] = s, y, ys
        d_ref = -G[r]
        if ys > cau:
            if end in order:
                order.remove(end)
            order.insert(0, end)
            order = order[:m]
            end = (end + 1) % m
            d_ref = two_loop(S, Y, YS, G[r], order) if k % direction_every == 0 else None
        rep["min_margin"] = min(rep["min_margin"], abs(ys - cau) / max(abs(cau), 1e-300))
        step = 1.0
        xp, gp_gpu = X[r].copy(), G[r].copy()
        e = r + 1
    return rep
