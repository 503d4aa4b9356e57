"""CPU: the lane-level logic of the step-parallel dvr kernels' consume phase (csrc/dvr_par_kernels.h `par_consume`:
chunks of 63 new steps with the still-open sample of the previous chunk on lane 0, run detection, the rewind
recurrence of a run of equal voxels iterated to its fixed point, commit flags -> sample index by prefix count, the
previous committed sample by "highest set bit below me", wave-uniform carries) restated with numpy over 64 "lanes" and
checked against the sequential integrator semantics of csrc/dvr_march.h (`Integrator::sample` / `commit`: dvxlr.cu:366-377
merge, W_{k-1} = T_{k-1} (d_k - d_{k-1}), P_k, k_surface) -- sample lists, dt bit for bit, sums to fp64 round-off.
What must be kept in step with the kernel: the chunk geometry (lane 0 = carry, lanes 1..63 = steps sb .. sb+62), the
commit rule (last valid lane commits only when the ray ends in this chunk; otherwise a lane commits iff its successor
is not the same voxel), `__shfl_down` past the last lane returning the lane's own value.  The GPU tests remain the
parity tests proper; this one catches logic slips (carry across chunks, runs that straddle a chunk) without a GPU."""
import numpy as np
import pytest


def seq_integrate(vids, ds, sig, merged, true_len):
    k = 0; csd = 0.0; dprev = 0.0; d0 = 0.0; S = 0.0
    pending = False; uvid = -1; ud = 0.0; udt = 0.0
    out = []
    fly = [0.0, 0.0]

    def commit(vid, d, dt):
        nonlocal k, csd, dprev, d0, S
        w = 0.0
        if k == 0:
            d0 = d
        else:
            csd = fly[0] * fly[1] if k == 1 else csd + fly[0] * fly[1]
            T = float(np.exp(np.float32(-csd)))
            w = T * (d - dprev); S += w
        out.append((k, vid, d, dt, S, w))
        fly[0] = float(sig[vid]); fly[1] = dt; dprev = d; k += 1
    last = 0.0
    for vid, d in zip(vids, ds):
        if merged:
            same = pending and vid == uvid
            if pending and not same:
                commit(uvid, ud, udt)
            udt = max(0.0, d - (last - (udt if same else 0.0))); ud = d; uvid = vid; pending = True
        else:
            commit(vid, d, max(0.0, d - last))
        last = d
    if merged and pending:
        commit(uvid, ud, udt)
    ks = next((kk for (kk, _, d, _, _, _) in out if d >= true_len), -1)
    return out, k, d0, S, dprev, ks


def par_consume(vids, ds, sig, merged, true_len):
    S = len(vids); lane = np.arange(64)
    k_base = 0; ksurf = 1 << 30; csd_c = 0.0; T_c = 1.0; dl_c = 0.0; d0 = 0.0; ssum = np.zeros(64); P_c = 0.0
    has_carry = False; c_vid = 0; c_d = 0.0; c_udt = 0.0
    out = {}
    for sb in range(0, S, 63):
        s = sb + lane - 1
        done = sb + 63 >= S
        last_lane = min(63, S - sb)
        valid = np.where(lane == 0, has_carry, s < S)
        vid = np.full(64, c_vid); d = np.full(64, c_d); lastd = np.zeros(64); udt = np.full(64, c_udt)
        for l in range(1, 64):
            if valid[l]:
                vid[l] = vids[s[l]]; d[l] = ds[s[l]]; lastd[l] = ds[s[l] - 1] if s[l] > 0 else 0.0
                udt[l] = max(0.0, d[l] - lastd[l])
        same = np.zeros(64, bool)
        if merged:
            for l in range(1, 64):
                same[l] = valid[l] and valid[l - 1] and vid[l] == vid[l - 1]
            while True:
                up = np.concatenate([[0.0], udt[:-1]])                     # wave_shr:1, lane 0 reads 0
                nu = np.where(same, np.maximum(0.0, d - (lastd - up)), udt)
                changed = (nu.view(np.int64) != udt.view(np.int64)).any(); udt = nu
                if not changed:
                    break
        same_dn = np.concatenate([same[1:], [False]]) if merged else np.zeros(64, bool)   # wave_shl:1, lane 63 reads 0
        commit = valid & np.where(lane == last_lane, done, ~same_dn)
        kl = k_base + np.array([commit[:l].sum() for l in range(64)])
        sg = np.where(commit, sig[vid], 0.0)
        csd = csd_c + np.cumsum(np.where(commit, sg.astype(np.float64) * udt, 0.0))
        T = np.exp((-csd).astype(np.float32)).astype(np.float64)
        Tp = np.zeros(64); dp = np.zeros(64)
        for l in range(64):
            below = np.nonzero(commit[:l])[0]
            Tp[l], dp[l] = (T[below[-1]], d[below[-1]]) if len(below) else (T_c, dl_c)
        w = np.where(commit & (kl > 0), Tp * (d - dp), 0.0)
        if commit.any() and k_base == 0:
            d0 = d[np.argmax(commit)]
        ssum += w
        P = P_c + np.cumsum(w)
        for l in np.nonzero(commit)[0]:
            out[kl[l]] = (kl[l], int(vid[l]), d[l], udt[l], P[l], w[l])
            if d[l] >= true_len:
                ksurf = min(ksurf, kl[l])
        P_c = P[63]; csd_c = csd[63]
        if commit.any():
            hl = np.max(np.nonzero(commit)[0]); T_c = T[hl]; dl_c = d[hl]; k_base += int(commit.sum())
        has_carry = not done
        if not done:
            c_vid, c_d, c_udt = vid[63], d[63], udt[63]
    return [out[k] for k in sorted(out)], k_base, d0, P_c, dl_c, (-1 if ksurf == (1 << 30) else ksurf)


@pytest.mark.parametrize("merged", [False, True])
def test_lane_level_consume_equals_sequential_integrator(merged):
    rng = np.random.default_rng(5 + merged)
    for trial in range(150):
        S = int(rng.integers(1, 300))
        vids = []
        while len(vids) < S:                     # runs of equal voxels, some longer than a whole 63-step chunk
            vids += [int(rng.integers(0, 50))] * (int(rng.choice([1, 1, 1, 2, 2, 3, 5, 70])) if merged else 1)
        vids = vids[:S]
        ds = np.cumsum(rng.uniform(0.0, 1.5, S) * (rng.uniform(size=S) > 0.1))      # non-decreasing, some zero-length steps
        sig = rng.uniform(0, 2, 50).astype(np.float32)
        tl = float(rng.uniform(0, ds[-1] * 1.2))
        a = seq_integrate(vids, ds, sig, merged, tl)
        b = par_consume(vids, ds, sig, merged, tl)
        assert a[1] == b[1] and a[5] == b[5] and a[2] == b[2] and a[4] == b[4], (trial, S)
        assert abs(a[3] - b[3]) <= 1e-9 * max(1.0, abs(a[3]))
        for x, y in zip(a[0], b[0]):
            assert x[:4] == y[:4], (trial, x, y)                        # k, voxel, d and dt: exact
            assert abs(x[4] - y[4]) <= 1e-9 * max(1.0, abs(x[4])) and abs(x[5] - y[5]) <= 1e-6 * max(1e-3, abs(x[5]))
