"""The device-resident detect path: workspace filter, candidate compaction, score write-back and
selectGrasps on the GPU (gpd_hip_detect / gpd_hip_detect_select), and the two-clouds-in-flight batch entry
(gpd_hip_detect_batch, BASELINE configs[4]) — against the oracle and against the unfused
search -> host filter -> images -> score route through the same C-ABI."""
import numpy as np
import pytest

from gpd_amd import api, synth

pytestmark = pytest.mark.gpu


def _weights(C=15):
    import os
    g = os.path.join(os.path.dirname(__file__), "golden", "lenet%d_params.npz" % C)
    return synth.lenet_weights(C, real=dict(np.load(g)) if os.path.exists(g) else None, trained_magnitude=True)  # |score| < 20: the range in which "within 1e-4" can be decided


@pytest.fixture(scope="module")
def ctx(lenet15_real):
    c = api.Context(api.default_params(15))
    c.set_lenet_weights(lenet15_real)
    yield c
    c.close()


def _oracle_candidates(oracle_mod, cl, si, w, p=None):
    p = p or oracle_mod.default_params(15)
    oh, on, _ = oracle_mod.detect(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, w)
    flat = oh.reshape(-1)
    return oh, flat[flat["valid"].astype(bool)].copy(), on


def _same_records(a, b):
    """byte for byte except the float score (asserted to 1e-4, observed identical)"""
    assert a.shape == b.shape
    assert np.abs(a["score"] - b["score"]).max() <= 1e-4 if len(a) else True
    x, y = a.copy(), b.copy()
    x["score"] = 0
    y["score"] = 0
    assert x.tobytes() == y.tobytes()


def test_fused_detect_equals_unfused_route(ctx, oracle_mod, cloud30k, lenet15_real):
    """gpd_hip_detect (filter + compaction + scatter on the device) against search -> oracle's
    filterGraspsWorkspace on the host -> images -> score through the same C-ABI."""
    cl = cloud30k
    si = synth.sample_indices(cl, 260)
    ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
    fused, n = ctx.detect(si)
    hands = oracle_mod.filter_workspace(oracle_mod.default_params(15), ctx.search(si))
    _, cand = ctx.images(hands, download=False)
    scores = ctx.score(None, n=len(cand))
    flat = hands.reshape(-1)
    flat["score"][cand] = scores
    assert n == len(cand) and n > 300
    assert fused.tobytes() == hands.tobytes()
    # the tight workspace really filters something on the device too
    p = api.default_params(15)
    p.workspace_grasps[:] = [-0.1, 0.1, -0.1, 0.12, -1.0, 1.0]
    op = oracle_mod.default_params(15)
    op.workspace_grasps[:] = [-0.1, 0.1, -0.1, 0.12, -1.0, 1.0]
    c2 = api.Context(p)
    try:
        c2.set_lenet_weights(lenet15_real)
        c2.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        f2, n2 = c2.detect(si)
        oh, on, _ = oracle_mod.detect(op, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, lenet15_real)
        assert n2 == on and 0 < n2 < n
        _same_records(f2.reshape(-1), oh.reshape(-1))
    finally:
        c2.close()


def test_detect_select_all_and_topk(ctx, oracle_mod, cloud30k, lenet15_real):
    cl = cloud30k
    si = synth.sample_indices(cl, 300)
    ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
    oh, ocand, on = _oracle_candidates(oracle_mod, cl, si, lenet15_real)
    # every candidate, in (set, slot) order
    got, ns, nc = ctx.detect_select(si, 0)
    assert ns == oh.shape[0] and nc == on == len(got) and nc > 400
    _same_records(got, ocand)
    # selectGrasps(k): the oracle's std::partial_sort restatement and, scores being distinct here, plain argsort
    assert len(np.unique(ocand["score"])) == len(ocand)
    for k in (1, 7, 100, 1000, nc, nc + 5):
        sel, _, nc2 = ctx.detect_select(si, k)
        want = oracle_mod.select(ocand["score"], k)
        assert nc2 == nc and len(sel) == min(k, nc)
        assert np.array_equal(want, np.argsort(-ocand["score"], kind="stable")[: len(want)])
        _same_records(sel, ocand[want])


def test_detect_select_with_tied_scores(oracle_mod, cloud30k, lenet15_real):
    """Equal scores: the reference's selectGrasps returns whatever arrangement std::partial_sort leaves
    (grasp_detector.cpp:409).  With ip2 = 0 every score is the same; with most ip2 weights zeroed and a
    ReLU-dead ip1 many are — both must come back in exactly that arrangement."""
    cl = cloud30k
    si = synth.sample_indices(cl, 200)
    for variant in ("all_equal", "many_equal"):
        w = {k: v.copy() for k, v in lenet15_real.items()}
        if variant == "all_equal":
            w["f2w"][:] = 0.0
        else:
            w["f1b"][:] = -1e6  # ReLU kills every ip1 unit ...
            w["f1b"][:3] = 0.0  # ... but three
            w["f1w"] = np.round(w["f1w"] * 8.0) / 8.0
            w["f2w"] = np.round(w["f2w"] * 4.0) / 4.0
        c = api.Context(api.default_params(15))
        try:
            c.set_lenet_weights(w)
            c.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
            allc, _, nc = c.detect_select(si, 0)
            _, ocand, on = _oracle_candidates(oracle_mod, cl, si, w)
            assert nc == on and np.array_equal(allc["score"], ocand["score"])
            if variant == "all_equal":
                assert len(np.unique(ocand["score"])) == 1
            for k in (5, 64, 300):
                sel, _, _ = c.detect_select(si, k)
                want = oracle_mod.select(ocand["score"], k)
                _same_records(sel, ocand[want])
        finally:
            c.close()


def test_detect_batch_equals_separate_calls(ctx, oracle_mod, lenet15_real):
    """configs[4] shape: independent clouds, two in flight on the context's two streams.  Results are those
    of separate upload + detect calls, whatever the cloud's lane, camera set-up or neighbours in the batch."""
    clouds, samples = [], []
    for cid in range(5):
        cl = dict(synth.make_cloud(300 + cid, 14000 + 1500 * cid))
        if cid == 2:  # two cameras, another view point: other shadows than its neighbours in the batch
            P = len(cl["xyz"])
            cam = np.zeros((2, P), np.int32)
            cam[0] = cl["xyz"][:, 0] < 0.05
            cam[1] = cl["xyz"][:, 0] > -0.05
            cl["cam_source"] = cam
            cl["view_points"] = np.array([[0.0, 0.0, 0.0], [0.3, 0.1, 0.05]])
        clouds.append(cl)
        samples.append(synth.sample_indices(cl, 90 + 10 * cid))
    samples[3] = samples[3][:0]  # a cloud without samples
    res = ctx.detect_batch(clouds, samples, 0)
    assert len(res) == 5
    for cid, (cl, si) in enumerate(zip(clouds, samples)):
        hands, ns, nc, ms = res[cid]
        ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        one, ns1, nc1 = ctx.detect_select(si, 0)
        assert (ns, nc) == (ns1, nc1) and hands.tobytes() == one.tobytes()
        if cid in (0, 2):
            _, ocand, on = _oracle_candidates(oracle_mod, cl, si, lenet15_real)
            assert on == nc and nc > 50
            _same_records(hands, ocand)
        if cid == 3:
            assert nc == 0 and len(hands) == 0
    # selectGrasps inside the batch, twice the same batch -> the same bytes (lanes reuse their buffers)
    a = ctx.detect_batch(clouds, samples, 25)
    b = ctx.detect_batch(clouds[::-1], samples[::-1], 25)[::-1]
    for cid in range(5):
        assert a[cid][0].tobytes() == b[cid][0].tobytes() and a[cid][1:3] == b[cid][1:3]
        want = res[cid][0][oracle_mod.select(res[cid][0]["score"], 25)] if len(res[cid][0]) else res[cid][0]
        assert a[cid][0].tobytes() == want.tobytes()


def test_detect_batch_with_tied_scores_and_repeated_samples(oracle_mod, lenet15_real):
    """Four clouds, selectGrasps inside the batch, equal scores among the winners (all scores equal; and sample indices
    drawn WITH repetition, as the reference's subsampleSampleIndices does — duplicated hand sets, duplicated scores).
    The std::partial_sort rerun of a job happens after the lane's search / plan buffers already hold the cloud after
    next: it must still return THIS cloud's records (it gathered from the wrong job's buffers)."""
    clouds, samples = [], []
    rng = np.random.RandomState(11)
    for cid in range(4):
        cl = synth.make_cloud(500 + cid, 12000 + 2000 * cid)
        si = synth.sample_indices(cl, 60)
        clouds.append(cl)
        samples.append(si[rng.randint(0, len(si), 90)].astype(np.int32))  # with repetition
    for variant in ("repeated_samples", "all_equal"):
        w = {k: v.copy() for k, v in lenet15_real.items()}
        if variant == "all_equal":
            w["f2w"][:] = 0.0
        c = api.Context(api.default_params(15))
        try:
            c.set_lenet_weights(w)
            res = c.detect_batch(clouds, samples, 40)
            for cid, (cl, si) in enumerate(zip(clouds, samples)):
                _, ocand, on = _oracle_candidates(oracle_mod, cl, si, w)
                assert res[cid][2] == on and on > 60
                if variant == "all_equal":  # (repeated samples repeat the hand sets, but each set draws its own shadow points)
                    assert len(np.unique(ocand["score"])) == 1
                _same_records(res[cid][0], ocand[oracle_mod.select(ocand["score"], 40)])
        finally:
            c.close()


def test_detect_select_beyond_the_device_selection_capacity(ctx, oracle_mod, cloud30k, lenet15_real):
    """num_selected larger than what select_topk_kernel sorts in LDS (8192): std::partial_sort on the downloaded scores,
    no limit — "all candidates, sorted" is a legitimate request (grasp_detector.cpp:405-420 has none)."""
    cl = cloud30k
    si = synth.sample_indices(cl, 4200)
    ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
    allc, _, nc = ctx.detect_select(si, 0)
    assert nc > 8192 + 500
    for k in (8193, nc, nc + 1000):
        sel, _, _ = ctx.detect_select(si, k)
        want = oracle_mod.select(allc["score"], k)
        assert len(sel) == min(k, nc) and sel.tobytes() == allc[want].tobytes()


def test_detect_batch_multi_two_contexts_on_one_device(ctx, lenet15_real):
    """gpd_hip_detect_batch_multi: one host thread per context, cloud i -> context i mod G.  Two contexts on the one
    GPU of the test box stand in for two GPUs; results are those of the single-context batch, cloud by cloud."""
    clouds = [synth.make_cloud(700 + cid, 10000 + 1000 * cid) for cid in range(7)]
    samples = [synth.sample_indices(cl, 50 + 5 * cid) for cid, cl in enumerate(clouds)]
    want = ctx.detect_batch(clouds, samples, 0)
    other = api.Context(api.default_params(15))
    try:
        other.set_lenet_weights(lenet15_real)
        for nsel in (0, 30):
            got = ctx.detect_batch_multi([other], clouds, samples, nsel)
            ref = want if nsel == 0 else ctx.detect_batch(clouds, samples, nsel)
            assert len(got) == len(clouds)
            for g, r in zip(got, ref):
                assert g[1:3] == r[1:3] and g[0].tobytes() == r[0].tobytes()
        with pytest.raises(api.GpdHipError, match="listed twice"):
            ctx.detect_batch_multi([ctx], clouds, samples, 0)
    finally:
        other.close()


def test_detect_batch_multi_three_contexts_uneven_jobs(ctx, lenet15_real):
    """Three contexts (stand-ins for three GPUs), seven and then two clouds: job i -> context i mod 3 gives shares of
    3 / 2 / 2 and 1 / 1 / 0 — an idle context, uneven lanes — and the results stay those of the single-context batch in
    cloud order.  Every worker thread binds itself to its device's NUMA node on the way (gpd_hip_bind_host_thread)."""
    clouds = [synth.make_cloud(800 + cid, 9000 + 700 * cid) for cid in range(7)]
    samples = [synth.sample_indices(cl, 40 + 7 * cid) for cid, cl in enumerate(clouds)]
    want = ctx.detect_batch(clouds, samples, 0)
    others = [api.Context(api.default_params(15)) for _ in range(2)]
    try:
        for o in others:
            o.set_lenet_weights(lenet15_real)
        for n in (7, 2, 1):
            got = ctx.detect_batch_multi(others, clouds[:n], samples[:n], 0)
            assert len(got) == n
            for g, r in zip(got, want[:n]):
                assert g[1:3] == r[1:3] and g[0].tobytes() == r[0].tobytes()
        node, ncpu = api.bind_host_thread(0)  # -1: the box exposes no topology; otherwise a non-empty CPU set
        assert node == -1 or ncpu > 0
    finally:
        for o in others:
            o.close()


def test_reserve_then_batch_allocates_nothing(lenet15_real, oracle_mod):
    """gpd_hip_reserve sizes both lanes once; a batch within those sizes then books no buffer growth on any cloud (`allocs`),
    its host timeline is ordered, and the results are those of the single-cloud entry.  Without the reservation the batch
    entry sizes the lanes itself: the growths are booked on the first cloud only."""
    clouds = [synth.make_cloud(900 + cid, 12000 + 900 * cid) for cid in range(5)]
    samples = [synth.sample_indices(cl, 60 + 9 * cid) for cid, cl in enumerate(clouds)]
    for reserve in (True, False):
        c = api.Context(api.default_params(15))
        try:
            c.set_lenet_weights(lenet15_real)
            if reserve:
                c.reserve(max_points=max(len(cl["xyz"]) for cl in clouds), max_cams=1, max_samples=max(len(s) for s in samples))
            got = c.detect_batch(clouds, samples, 0)
            tl = c.last_batch_timeline
            allocs = [a for _, a in tl]
            # cloud 1 holds one neighbourhood of 8243 points: its search is re-run with the next list capacity, which re-allocates
            # the lane's lists — once, keeping the sample capacity the lane was sized for (cloud 3, same lane, more samples: 0)
            if reserve:
                assert allocs == [0, 1, 0, 0, 0], allocs
            else:
                assert allocs[0] > 0 and allocs[1:] == [1, 0, 0, 0], allocs
            for h, _ in tl:
                assert 0 < h[0] <= h[2] and h[1] <= h[2] <= h[3] <= h[4], h
            c.detect_batch(clouds, samples, 0)  # a second batch of the same sizes: nothing grows either way
            assert sum(a for _, a in c.last_batch_timeline) == 0
            p = oracle_mod.default_params(15)
            for (hands, ns, nc, _), cl, si in zip(got, clouds, samples):
                oh, on, _ = oracle_mod.detect(p, cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"], si, lenet15_real)
                flat = oh.reshape(-1)
                want = flat[np.flatnonzero(flat["valid"])]
                assert nc == on and len(hands) == len(want)
                a, b = hands.copy(), want.copy()
                assert np.abs(a["score"] - b["score"]).max() <= 1e-4
                a["score"] = 0
                b["score"] = 0
                assert a.tobytes() == b.tobytes()
        finally:
            c.close()
    with pytest.raises(api.GpdHipError, match="bad argument"):
        c2 = api.Context(api.default_params(15))
        try:
            c2.reserve(max_points=0)
        finally:
            c2.close()


def test_detect_batch_reports_a_bad_cloud(ctx, cloud30k):
    cl = cloud30k
    si = synth.sample_indices(cl, 40)
    bad = dict(cl)
    bad["xyz"] = cl["xyz"].copy()
    bad["xyz"][17, 1] = np.inf
    with pytest.raises(api.GpdHipError, match="non-finite"):
        ctx.detect_batch([cl, bad, cl], [si, si, si], 0)
    with pytest.raises(api.GpdHipError, match="non-finite"):
        ctx.upload_cloud(bad["xyz"], bad["normals"], bad["cam_source"], bad["view_points"])
    # the context is still usable
    ctx.upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
    _, n = ctx.detect(si)
    assert n > 20


def test_create_rejects_bad_parameters():
    for field, value in (("hand_axes", 3), ("volume_width", 0.0), ("hand_depth", -0.06), ("finger_width", float("nan")),
                         ("hand_depth", 0.9)):  # 178 deepening steps: beyond the 128-entry table
        p = api.default_params(15)
        if field == "hand_axes":
            p.hand_axes[0] = value
        else:
            setattr(p, field, value)
        with pytest.raises(api.GpdHipError):
            api.Context(p)


def test_one_cloud_sharded_by_sample_range_equals_the_single_call(cloud30k, lenet15_real):
    """gpd_hip_detect_sharded: the samples of ONE cloud cut into contiguous ranges over 2 and 3 contexts (here on one device).
    The reference draws a cloud's shadow points from one LCG stream, set after set (hand_set.cpp:268-283); the ranges' draw totals
    are scanned on the host and every range starts its stream where the ones before it stopped — so the concatenated records,
    scores included (they see every image byte, shadow channels included), are byte for byte those of the single-context call,
    for even, uneven and empty shares; without the scan they are not."""
    si = synth.sample_indices(cloud30k, 700)
    one = api.Context(api.default_params(15))
    others = [api.Context(api.default_params(15)) for _ in range(2)]
    try:
        for c in [one] + others:
            c.set_lenet_weights(lenet15_real)
        one.upload_cloud(cloud30k["xyz"], cloud30k["normals"], cloud30k["cam_source"], cloud30k["view_points"])
        want, ns, nc = one.detect_select(si, 0)
        assert nc > 1200
        for ctxs, split in (((one, others[0]), None), ((one, others[0]), [123]), ((one, others[0], others[1]), None),
                            ((others[1], one, others[0]), [1, 650]), ((one, others[0], others[1]), [0, 700])):
            got, info = ctxs[0].detect_sharded(ctxs[1:], cloud30k, si, split)
            assert sum(i[1] for i in info) == nc and sum(i[0] for i in info) == ns
            assert info[0][2] == 0 and all(info[g][2] == sum(i[3] for i in info[:g]) for g in range(len(info)))
            assert got.tobytes() == want.tobytes(), (len(ctxs), split)
        # the scan matters: the second half on its own (its stream restarted) scores differently
        half = len(si) // 2
        alone, _, _ = others[0].detect_batch([cloud30k], [si[half:]], 0)[0][:3]
        tail = want[len(want) - len(alone):]
        assert len(alone) > 500 and np.array_equal(alone["position"], tail["position"]) and not np.array_equal(alone["score"], tail["score"])
    finally:
        one.close()
        for c in others:
            c.close()


def test_sharded_dense_cloud(lenet15_real):
    """BASELINE configs[3] across contexts: a 300k-point clutter cloud, 1500 samples, three ranges."""
    cl = synth.make_cloud(1234, 300000, clutter=True)
    si = synth.sample_indices(cl, 1500)
    ctxs = [api.Context(api.default_params(15)) for _ in range(3)]
    try:
        for c in ctxs:
            c.set_lenet_weights(lenet15_real)
        ctxs[0].upload_cloud(cl["xyz"], cl["normals"], cl["cam_source"], cl["view_points"])
        want, ns, nc = ctxs[0].detect_select(si, 0)
        got, info = ctxs[0].detect_sharded(ctxs[1:], cl, si)
        assert nc > 2000 and sum(i[1] for i in info) == nc and got.tobytes() == want.tobytes()
    finally:
        for c in ctxs:
            c.close()
