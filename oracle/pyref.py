"""Independent Python restatement of the hot path (TEST INFRASTRUCTURE ONLY — see oracle/smpc_oracle.cpp header).

Purpose: a second, independently written statement of the same maths to cross-check the C++ oracle, because the
reference has no tests / golden vectors and Ceres is absent (PARITY UNPINNED):
  * residuals written with torch.float64 tensor ops straight from the reference formulas
    (include/nav2_social_mpc_controller/critics/*_cost_function.hpp, update_state.hpp); the Jacobian comes from
    torch.autograd (reverse mode) — a different differentiation mechanism than the oracle's forward duals and
    the HIP kernels' analytic gradients;
  * the trust-region LM loop written again from SURVEY.md Appendix A with numpy linear algebra
    (np.linalg.lstsq for DENSE_QR-like steps, np.roots = companion-matrix eigenvalues for the line-search
    polynomial, like Ceres does).
It is slow (python loops) and only used on a handful of small scenes.
"""
import math

import numpy as np
import torch

torch.set_default_dtype(torch.float64)
torch.set_num_threads(1)  # tiny scalar ops: threading only adds overhead

LAMBDA, GAMMA, NPRIME, NN, FORCE_FACTOR = 2.0, 0.35, 3.0, 2.0, 2.1   # src/critics/social_work_cost_function.cpp:38-43
ALPHA, D0 = 3.0, 0.5                                                  # src/critics/proxemics_cost_function.cpp:37-38


def _wrap_to_pi(a):
    # critics/social_work_cost_function.hpp:39-46 ; the shift is a constant wrt differentiation
    v = float(a.detach())
    shift = 0.0
    while v + shift > math.pi:
        shift -= 2.0 * math.pi
    while v + shift <= -math.pi:
        shift += 2.0 * math.pi
    return a + shift


def _social_force(me_pos, me_vel, other_pos, other_vel):
    """computeSocialForce for ONE other agent (critics/social_work_cost_function.hpp:173-225)."""
    diff = me_pos - other_pos
    if float(torch.linalg.norm(diff.detach())) < 1e-6:
        diff = torch.tensor([1e-6, 0.0])
    dist = torch.sqrt(diff[0] ** 2 + diff[1] ** 2)
    direction = diff / dist
    vel_diff = me_vel - other_vel
    iv = LAMBDA * vel_diff + direction
    il = torch.sqrt(iv[0] ** 2 + iv[1] ** 2)
    idir = iv / il
    theta = _wrap_to_pi(torch.atan2(direction[1], direction[0]) - torch.atan2(idir[1], idir[0]))
    B = GAMMA * il
    fv = -torch.exp(-dist / B - (NPRIME * B * theta) ** 2)
    sign = 1.0 if float(theta.detach()) > 0 else -1.0
    fa = -sign * torch.exp(-dist / B - (NN * B * theta) ** 2)
    left = torch.stack([-idir[1], idir[0]])
    return FORCE_FACTOR * (fv * idir + fa * left)


def _bicubic(costmap, r, c):
    """Catmull-Rom bicubic with clamp-to-edge (Ceres BiCubicInterpolator<Grid2D<u8>>)."""
    size_y, size_x = costmap.shape
    row, col = int(math.floor(float(r.detach()))), int(math.floor(float(c.detach())))

    def val(rr, cc):
        rr = min(max(rr, 0), size_y - 1)
        cc = min(max(cc, 0), size_x - 1)
        return float(costmap[rr, cc])

    def spline(p0, p1, p2, p3, x):
        a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3)
        b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3)
        cc = 0.5 * (-p0 + p2)
        return p1 + x * (cc + x * (b + x * a))

    xc = c - col
    rows = [spline(val(row - 1 + i, col - 1), val(row - 1 + i, col), val(row - 1 + i, col + 1),
                   val(row - 1 + i, col + 2), xc) for i in range(4)]
    return spline(rows[0], rows[1], rows[2], rows[3], r - row)


def residuals(prm, sc, b, x):
    """Residual vector (torch, differentiable wrt x) of scene b in the reference's order (src/optimizer.cpp:251-371)."""
    T, N, dt = sc.T, sc.N, sc.dt
    CH, bl, nb, P, M, nbounded = prm.dims(T, bool(sc.has_people[b]))
    x0, y0, yaw0 = (float(v) for v in sc.pose0[b])
    blast = (CH - 1) // bl
    blk = lambda j: (j // bl) if j < CH else blast
    # shared rollout: pose after k steps (update_state.hpp:46-61)
    xs, ys, ths = [torch.tensor(x0)], [torch.tensor(y0)], [torch.tensor(yaw0)]
    for j in range(T):
        v, w = x[2 * blk(j)], x[2 * blk(j) + 1]
        xs.append(xs[-1] + v * torch.cos(ths[-1]) * dt)
        ys.append(ys[-1] + v * torch.sin(ths[-1]) * dt)
        ths.append(ths[-1] + w * dt)
    gx, gy = float(sc.path_pts[b, T, 0]), float(sc.path_pts[b, T, 1])
    cm = sc.costmap[0 if sc.costmap_shared else b]
    ox, oy = (float(v) for v in sc.costmap_origin[0 if sc.costmap_shared else b])
    out = []
    for i in range(T):
        px, py, th = xs[i + 1], ys[i + 1], ths[i + 1]
        if sc.has_people[b]:
            ppl = sc.people[b, i + 1]  # [6, N]
            # --- AgentAngle (critics/agent_angle_cost_function.hpp:125-195)
            closest, best = -1, math.inf
            for a in range(N):
                d2 = (ppl[0, a] - x0) ** 2 + (ppl[1, a] - y0) ** 2
                if d2 < best and ppl[4, a] > 0.05:
                    best, closest = d2, a
            r_aa = torch.tensor(0.0)
            if closest >= 0 and best <= 4.0:
                ang0 = math.atan2(ppl[1, closest] - y0, ppl[0, closest] - x0)
                hd = math.atan2(math.sin(ppl[2, closest] - yaw0), math.cos(ppl[2, closest] - yaw0))
                rel = math.atan2(math.sin(ang0 - yaw0), math.cos(ang0 - yaw0))
                target = None
                if hd <= -5 * math.pi / 6 or hd >= math.pi / 6:
                    if not rel < 0.0:
                        target = yaw0 - math.pi / 6
                else:
                    if not rel > 0.0:
                        target = yaw0 + math.pi / 6
                if target is not None:
                    ad = torch.atan2(torch.sin(th - target), torch.cos(th - target))
                    r_aa = prm.agent_angle_weight * ad * ad
            out.append(r_aa)
            # --- SocialWork (critics/social_work_cost_function.hpp:102-150)
            v = x[2 * blk(i)]
            rpos = torch.stack([px, py])
            rvel = torch.stack([v * torch.cos(th), v * torch.sin(th)])
            fr = torch.zeros(2)
            wp = torch.tensor(0.0)
            for a in range(N):
                apos = torch.tensor([ppl[0, a], ppl[1, a]])
                avel = torch.tensor([ppl[4, a] * math.cos(ppl[2, a]), ppl[4, a] * math.sin(ppl[2, a])])
                if ppl[3, a] != -1.0:
                    fr = fr + _social_force(rpos, rvel, apos, avel)
                fa = _social_force(apos, avel, rpos, rvel)   # every column, valid or not (:137-143)
                wp = wp + fa[0] ** 2 + fa[1] ** 2
            out.append(prm.social_weight * (fr[0] ** 2 + fr[1] ** 2 + wp + 1e-6))
            # --- Proxemics (critics/proxemics_cost_function.hpp:125-151)
            mind, arg = None, -1
            for a in range(N):
                if ppl[3, a] == -1.0:
                    continue
                d2 = (px - ppl[0, a]) ** 2 + (py - ppl[1, a]) ** 2
                if mind is None or float(d2.detach()) < float(mind.detach()):
                    mind, arg = d2, a
            if mind is None:
                out.append(torch.tensor(0.0) * x[0])
            else:
                out.append(prm.proxemics_weight * ALPHA * torch.exp(-mind / (D0 * D0)))
        # --- Velocity (critics/velocity_cost_function.hpp:89-99)
        if i < CH:
            out.append(prm.velocity_weight * (prm.desired_linear_vel - x[2 * (i // bl)]) ** 2)
        else:
            out.append(torch.tensor(0.0) * x[0])
        # --- GoalAlign (critics/goal_align_cost_function.hpp:100-116)
        g = float(sc.goal_yaw[b])
        ta = torch.atan2(torch.sin(g - th), torch.cos(g - th))
        out.append(prm.goal_align_weight * ta * ta)
        # --- Distance x2 (critics/distance_cost_function.hpp:117-132)
        q = (px - gx) ** 2 + (py - gy) ** 2
        out.append(prm.distance_weight * q * q)
        tx, ty = float(sc.path_pts[b, i + 1, 0]), float(sc.path_pts[b, i + 1, 1])
        q = (px - tx) ** 2 + (py - ty) ** 2
        out.append(prm.angle_weight * q * q)
        # --- Obstacle (critics/obstacle_cost_function.hpp:137-167)
        fx = px + 0.25 * torch.cos(th)
        fy = py + 0.25 * torch.sin(th)
        out.append(prm.obstacle_weight * _bicubic(cm, (fy - oy) / sc.resolution, (fx - ox) / sc.resolution))
        # --- VelocityFeasibility (critics/velocity_feasibility_cost_function.hpp:86-98; src/optimizer.cpp:364-370)
        if i != 0 and i < CH // bl:
            w = prm.velocity_feasibility_weight
            out.append(w * (x[2 * i] - x[2 * i - 2]) ** 2 + w * (x[2 * i + 1] - x[2 * i - 1]) ** 2)
    return torch.stack(out)


def evaluate(prm, sc, b, x, jacobian=True):
    xt = torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=jacobian)
    r = residuals(prm, sc, b, xt)
    if not jacobian:
        return r.detach().numpy(), None
    # one batched reverse pass: rows of the identity as grad_outputs
    J, = torch.autograd.grad(r, xt, grad_outputs=torch.eye(r.shape[0]), is_grads_batched=True, allow_unused=True)
    if J is None:
        J = torch.zeros((r.shape[0], xt.shape[0]))
    return r.detach().numpy(), J.numpy()


# ---------------------------------------------------------------------------------------------------------------
# Trust-region LM restated from SURVEY.md Appendix A (numpy)
# ---------------------------------------------------------------------------------------------------------------
def _interp_min(samples, lo, hi):
    """samples: list of (x, value, gradient or None). Minimise the interpolating polynomial on [lo, hi] (A.8)."""
    rows, rhs = [], []
    nc = sum(1 + (s[2] is not None) for s in samples)
    deg = nc - 1
    for (xv, val, grad) in samples:
        rows.append([xv ** (deg - j) for j in range(deg + 1)])
        rhs.append(val)
        if grad is not None:
            rows.append([(deg - j) * xv ** (deg - j - 1) if j < deg else 0.0 for j in range(deg + 1)])
            rhs.append(grad)
    poly = np.linalg.solve(np.array(rows), np.array(rhs))
    cand = [(lo + hi) / 2.0, lo, hi]
    best_x, best_v = cand[0], np.polyval(poly, cand[0])
    for c in cand[1:]:
        v = np.polyval(poly, c)
        if v < best_v:
            best_x, best_v = c, v
    if nc > 2:
        for root in np.roots(np.polyder(poly)):
            rr = float(np.real(root))
            if lo <= rr <= hi:
                v = np.polyval(poly, rr)
                if v < best_v:
                    best_x, best_v = rr, v
    for (xv, _, _) in samples:
        if lo <= xv <= hi:
            v = np.polyval(poly, xv)
            if v < best_v:
                best_x, best_v = xv, v
    return best_x


def solve(prm, sc, b, max_iterations=None):
    T = sc.T
    CH, bl, nb, P, M, nbounded = prm.dims(T, bool(sc.has_people[b]))
    lo = np.full(P, -np.inf)
    hi = np.full(P, np.inf)
    for k in range(min(nbounded, nb)):
        lo[2 * k], hi[2 * k], lo[2 * k + 1], hi[2 * k + 1] = prm.v_min, prm.v_max, prm.w_min, prm.w_max
    plus = lambda xx, d: np.minimum(np.maximum(xx + d, lo), hi)
    max_it = prm.max_iterations if max_iterations is None else max_iterations

    def ev(xx, jac):
        r, J = evaluate(prm, sc, b, xx, jacobian=jac)
        return 0.5 * float(r @ r), r, J

    x = plus(np.array(sc.init_params[b], dtype=np.float64), 0.0)
    cost, r, J = ev(x, True)
    g = J.T @ r
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(axis=0)))
    Js = J * scale
    gmax = np.max(np.abs(x - plus(x, -g)))
    radius, dec = 1e4, 2.0
    it, ninvalid, successful = 0, 0, True
    reason = "max_iterations"
    log = [(0, cost)]
    while True:
        if it >= max_it:
            reason = "max_iterations"
            break
        if successful and gmax <= prm.gradient_tol:
            reason = "gradient_tol"
            break
        if radius <= 1e-32:
            reason = "min_radius"
            break
        it += 1
        successful = False
        diag = np.clip((Js * Js).sum(axis=0), 1e-6, 1e32)
        D = np.sqrt(diag / radius)
        A = np.vstack([Js, np.diag(D)])
        rhs = np.concatenate([r, np.zeros(P)])
        y = np.linalg.lstsq(A, rhs, rcond=None)[0]
        step = -y
        mr = Js @ step
        mcc = -float(mr @ (r + mr / 2.0))
        if not (np.all(np.isfinite(step)) and mcc > 0.0):
            ninvalid += 1
            if ninvalid >= 5:
                reason = "invalid_steps"
                break
            radius /= dec
            dec *= 2.0
            continue
        ninvalid = 0
        delta = step * scale
        # projected Armijo line search
        gd0 = float(g @ delta)
        dirmax = np.max(np.abs(delta))

        def sample(alpha):
            c_, r_, J_ = ev(plus(x, alpha * delta), True)
            return (alpha, c_, float(delta @ (J_.T @ r_)))

        cur, prev, iters, ok = sample(1.0), None, 0, False
        while True:
            if cur[1] <= cost + 1e-4 * gd0 * cur[0]:
                ok = True
                break
            iters += 1
            if iters >= 20:
                break
            smp = [(0.0, cost, gd0), cur] + ([prev] if prev is not None else [])
            a = _interp_min(smp, 1e-3 * cur[0], 0.6 * cur[0])
            if a * dirmax < 1e-9:
                break
            prev, cur = cur, sample(a)
        if ok:
            delta = delta * cur[0]
        cand = plus(x, delta)
        ccost, _, _ = ev(cand, False)
        snorm = np.linalg.norm(x - cand)
        if snorm <= prm.param_tol * (np.linalg.norm(x) + prm.param_tol):
            reason = "parameter_tol"
            break
        if abs(cost - ccost) <= prm.fn_tol * cost:
            reason = "function_tol"
            break
        rho = (cost - ccost) / mcc
        if rho > 1e-3:
            x = cand
            cost, r, J = ev(x, True)
            g = J.T @ r
            Js = J * scale
            gmax = np.max(np.abs(x - plus(x, -g)))
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            dec = 2.0
            successful = True
        else:
            radius /= dec
            dec *= 2.0
        log.append((it, cost))
    return {"x": x, "cost": cost, "iterations": it, "reason": reason, "log": log}
