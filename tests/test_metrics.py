"""CPU: the evaluation arithmetic — ddpm_torch.metrics against fixture G15, written by the reference's own
functions (tests/golden/make_golden.py g15: metrics/fid_score.py:109-139,264-316 and metrics/precision_recall.py:159-206 on synthetic
features), plus the properties the formulas must have.  The pretrained feature networks are not available offline: the extractor is
pluggable and the constructors raise without one."""
import numpy as np
import pytest
import torch

import ddpm_torch
from ddpm_torch import metrics as M
from tests.golden.recipes import g15_inputs


@pytest.fixture(scope="module")
def g15(golden):
    return golden("g15_metrics.pt")


def _stats(x, sizes, D):
    st = M.InceptionStatistics(activation_dim=D, feature_extractor=lambda v: v)
    o = 0
    for n in sizes:
        st(x[o:o + n].reshape(n, D, 1, 1))          # un-pooled [N, D, 1, 1] maps, as the reference's network returns them
        o += n
    return st.get_statistics()


def test_streaming_statistics_and_frechet_distance_match_the_reference(g15):
    D = g15["D"]
    gen, real = g15_inputs(D, g15["seed"])
    assert float(gen.double().sum()) == pytest.approx(float(g15["gen_sum"]), rel=1e-12)      # the recipe recreated the reference's inputs
    mu_g, cov_g = _stats(gen, g15["gen_batches"], D)
    mu_r, cov_r = _stats(real, g15["real_batches"], D)
    for got, want in ((mu_g, g15["mu_g"]), (cov_g, g15["cov_g"]), (mu_r, g15["mu_r"]), (cov_r, g15["cov_r"])):
        assert np.allclose(got, want.numpy(), rtol=1e-9, atol=1e-12)
    assert M.calc_fd(mu_g, cov_g, mu_r, cov_r) == pytest.approx(g15["fd"], rel=1e-8)
    assert M.calc_fd(g15["mu_g"].numpy(), g15["cov_g"].numpy(), g15["mu_r"].numpy(), g15["cov_r"].numpy()) == pytest.approx(g15["fd"], rel=1e-8)
    assert abs(M.calc_fd(mu_g, cov_g, mu_g, cov_g)) < 1e-8
    # any split of the same samples gives the same statistics; one batch == the plain unbiased estimates
    mu1, cov1 = _stats(gen, (700,), D)
    mu7, cov7 = _stats(gen, (1, 99, 250, 350), D)
    assert np.allclose(mu1, mu7, atol=1e-12) and np.allclose(cov1, cov7, atol=1e-12)
    assert np.allclose(cov1, np.cov(gen.double().numpy(), rowvar=False), atol=1e-12) and np.allclose(mu1, gen.double().mean(0).numpy())
    # the distance is symmetric, grows with a mean shift by exactly |shift|^2, and survives rank-deficient covariances
    assert M.calc_fd(mu_r, cov_r, mu_g, cov_g) == pytest.approx(g15["fd"], rel=1e-8)
    shift = np.full(D, 0.25)
    assert M.calc_fd(mu_g + shift, cov_g, mu_g, cov_g) == pytest.approx(D * 0.0625, rel=1e-8)
    few = gen[:10]                                                 # 10 samples in 48 dimensions: singular covariance
    mu_f, cov_f = _stats(few, (10,), D)
    assert np.isfinite(M.calc_fd(mu_f, cov_f, mu_r, cov_r))


def test_manifold_radii_and_precision_recall_match_the_reference(g15):
    gen, real = g15_inputs(g15["D"], g15["seed"])
    fa, fb = gen[:400].contiguous(), real[:500].contiguous() * 0.9

    def builder(f, k):
        b = object.__new__(M.ManifoldBuilder)                      # fp32 features, like the fixture (the constructor stores fp16)
        b.nhood_size, b.row_batch_size, b.col_batch_size, b.op_device = k, 128, 200, torch.device("cpu")
        return b
    ka, kb = builder(fa, 3).compute_kth(fa), builder(fb, 3).compute_kth(fb)
    assert torch.equal(ka, g15["kth_a"]) and torch.equal(kb, g15["kth_b"]) and torch.equal(builder(fa, 5).compute_kth(fa), g15["kth_a5"])
    p, r = M.calc_pr(M.Manifold(fa, ka), M.Manifold(fb, kb), row_batch_size=128, col_batch_size=200, device=torch.device("cpu"))
    assert float(p) == pytest.approx(g15["precision"], abs=1e-7) and float(r) == pytest.approx(g15["recall"], abs=1e-7)
    # block sizes are an implementation detail
    p2, r2 = M.calc_pr(M.Manifold(fa, ka), M.Manifold(fb, kb), row_batch_size=37, col_batch_size=1000, device=torch.device("cpu"))
    assert float(p2) == float(p) and float(r2) == float(r)
    # a set against itself is fully covered; a far-away set not at all
    mb = M.ManifoldBuilder(features=fa, nhood_size=3, row_batch_size=128, col_batch_size=200)
    assert mb.features.dtype == torch.float16 and mb.kth.dtype == torch.float16 and mb.kth.shape == (400,)
    ps, rs = M.calc_pr(mb.manifold, mb.manifold, 128, 200, torch.device("cpu"))
    assert float(ps) == 1.0 and float(rs) == 1.0
    far = M.ManifoldBuilder(features=fa + 100.0, nhood_size=3, row_batch_size=128, col_batch_size=200)
    pf, rf = M.calc_pr(far.manifold, mb.manifold, 128, 200, torch.device("cpu"))
    assert float(pf) == 0.0 and float(rf) == 0.0
    assert torch.equal(M.precision_recall.to_uint8(g15["to_uint8_in"]), g15["to_uint8_out"])


def test_manifold_builder_extracts_features_from_every_kind_of_source(tmp_path):
    imgs = torch.randint(0, 256, (50, 3, 8, 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    extract = lambda x: x.float().mean(dim=(2, 3)) / 255.0                     # a stand-in "network": per-channel means
    a = M.ManifoldBuilder(data=imgs, extractor=extract, extr_batch_size=16, nhood_size=2)
    assert a.features.shape == (50, 3)
    torch.save(imgs, tmp_path / "imgs.pt")
    b = M.ManifoldBuilder(data=str(tmp_path / "imgs.pt"), extractor=extract, extr_batch_size=7, nhood_size=2)
    assert torch.equal(a.features, b.features) and torch.equal(a.kth, b.kth)
    c = M.ManifoldBuilder(data=torch.utils.data.TensorDataset(imgs), extractor=extract, extr_batch_size=20, nhood_size=2)
    assert torch.equal(a.features, c.features)
    sub = M.ManifoldBuilder(data=imgs, extractor=extract, max_sample_size=20, nhood_size=2)      # seeded subsample, like the reference
    np.random.seed(1234)
    assert torch.equal(sub.features, extract(imgs[torch.as_tensor(np.random.choice(50, size=20, replace=False))]).half())

    class Model:
        def sample_x(self, n):
            return torch.rand(n, 3, 8, 8) * 2 - 1
    m = M.ManifoldBuilder(model=Model(), extractor=extract, extr_batch_size=16, max_sample_size=40, nhood_size=2)
    assert m.features.shape == (40, 3)
    with pytest.raises(RuntimeError, match="extractor"):
        M.ManifoldBuilder(data=imgs)
    with pytest.raises(AssertionError, match="uint8"):
        M.ManifoldBuilder(data=imgs.float(), extractor=extract)


def test_evaluator_runs_the_reference_protocol_with_a_pluggable_extractor(tmp_path, monkeypatch):
    D = 6
    extract = lambda x: x.reshape(x.shape[0], -1)[:, :D]
    target = (np.zeros(D), np.eye(D))
    calls = []

    def sample_fn(sample_size, diffusion):
        calls.append((sample_size, diffusion))
        return torch.randn(sample_size, 3, 2, 2, generator=torch.Generator().manual_seed(len(calls)))
    ev = ddpm_torch.Evaluator("cifar10", diffusion="proc", eval_batch_size=64, eval_total_size=200, feature_extractor=extract, target_stats=target)
    ev.istats.activation_dim, ev.istats.running_mean, ev.istats.running_var = D, np.zeros(D), np.zeros((D, D))
    out = ev.eval(sample_fn)
    assert [c[0] for c in calls] == [64, 64, 64, 8] and all(c[1] == "proc" for c in calls)            # every batch, the remainder last
    assert ev.istats.count == 200 and 0 <= out["fid"] < 1.0                                            # N(0, I) samples against N(0, I)
    calls.clear()
    assert ev.eval(sample_fn, is_leader=False) == {"fid": None} and len(calls) == 4                    # non-leaders sample, do not score
    calls.clear()
    ev2 = ddpm_torch.Evaluator("cifar10", eval_batch_size=50, eval_total_size=100, feature_extractor=extract, target_stats=target)
    ev2.istats.activation_dim, ev2.istats.running_mean, ev2.istats.running_var = D, np.zeros(D), np.zeros((D, D))
    ev2.eval(sample_fn)
    assert [c[0] for c in calls] == [50, 50]                                                           # (the reference would ask for 0 last)
    # statistics file: offline loader with the reference's file name and keys
    np.savez(tmp_path / "fid_stats_cifar10_train.npz", mu=np.ones(D), sigma=2 * np.eye(D))
    mu, sigma = M.get_precomputed("cifar10", str(tmp_path))
    assert np.allclose(mu, 1) and np.allclose(sigma, 2 * np.eye(D))
    with pytest.raises(FileNotFoundError, match="fid_stats_celeba"):
        M.get_precomputed("celeba", str(tmp_path))
    # no extractor, no file named by the environment: loud failure, no substitute network
    monkeypatch.delenv("DDPM_TORCH_AMD_INCEPTION", raising=False)
    with pytest.raises(RuntimeError, match="feature network"):
        ddpm_torch.Evaluator("cifar10", target_stats=target)


def test_eval_cli_scores_a_sample_folder(tmp_path, monkeypatch):
    """``eval.py`` end to end (reference: eval.py:72-141) with stand-in feature networks: a folder of PNG samples against a dataset read from
    its own on-disk format; statistics / manifold computed from the raw data on the first run, loaded from ``--precomputed-dir`` on the
    second; results appended to metrics.txt; refused without a network."""
    import importlib.util
    import os
    import pickle
    from PIL import Image
    spec = importlib.util.spec_from_file_location("eval_cli", os.path.join(os.path.dirname(ddpm_torch.__file__), "..", "eval.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    rng = np.random.RandomState(0)
    folder = tmp_path / "data" / "cifar-10-batches-py"
    folder.mkdir(parents=True)
    real = []
    for fn in [f"data_batch_{i}" for i in range(1, 6)]:
        arr = rng.randint(0, 256, (12, 3072), dtype=np.uint8)
        real.append(arr.reshape(-1, 3, 32, 32))
        with open(folder / fn, "wb") as f:
            pickle.dump({"data": arr, "labels": [0] * 12}, f, protocol=2)
    real = torch.from_numpy(np.concatenate(real))
    gen = torch.from_numpy(np.concatenate([rng.randint(0, 256, (20, 32, 32, 3), dtype=np.uint8),       # half like the data,
                                           rng.randint(0, 160, (20, 32, 32, 3), dtype=np.uint8)]))     # half darker: 0 < precision < 1
    run = tmp_path / "images" / "run1"
    run.mkdir(parents=True)
    for i, g in enumerate(gen):
        Image.fromarray(g.numpy()).save(run / f"{i:03d}.png")
    D = 5
    fid_net = lambda x: torch.stack([x[:, 0].mean(dim=(1, 2)), x[:, 1].mean(dim=(1, 2)), x[:, 2].mean(dim=(1, 2)), x[:, :, :16].mean(dim=(1, 2, 3)),
                                     x[:, :, :, :16].mean(dim=(1, 2, 3))], dim=1) * 10     # noqa: E731  ([-1, 1] floats in)
    pr_net = lambda x: x.float().reshape(x.shape[0], 3, 4, 8, 4, 8).mean(dim=(3, 5)).reshape(x.shape[0], -1) / 16      # noqa: E731  (uint8 in)
    monkeypatch.setattr(M.InceptionStatistics.__init__, "__defaults__", (None, D, torch.device("cpu"), None))
    argv = ["--root", str(tmp_path / "data"), "--dataset", "cifar10", "--sample-folder", str(run) + "/", "--device", "cpu", "--num-workers", "0",
            "--eval-batch-size", "16", "--eval-total-size", "1000", "--precomputed-dir", str(tmp_path / "pre"), "--nhood-size", "2"]
    out = cli.main(argv, extractors={"fid": fid_net, "pr": pr_net})
    # the same numbers from the library, directly
    to_float = lambda u: (u.float() - 127.5) / 127.5                                      # noqa: E731
    sa, sb = M.InceptionStatistics(activation_dim=D, feature_extractor=fid_net), M.InceptionStatistics(activation_dim=D, feature_extractor=fid_net)
    sa(to_float(real)); sb(to_float(gen.permute(0, 3, 1, 2)))
    want_fid = M.calc_fd(*sb.get_statistics(), *sa.get_statistics())
    assert out["folder_name"] == "run1" and abs(out["fid"] - want_fid) <= 1e-6 * max(1.0, want_fid) and out["fid"] > 0.1     # (batches of 16 vs one: fp64 merge order)
    mt, mg = (M.ManifoldBuilder(features=pr_net(v), nhood_size=2).manifold for v in (real, gen.permute(0, 3, 1, 2)))
    p, r = M.calc_pr(mg, mt, 10000, 10000, torch.device("cpu"))
    assert out["pr"] == f"{float(p):.3f}/{float(r):.3f}" and 0.0 < float(p) < 1.0
    assert (tmp_path / "pre" / "fid_stats_cifar10.npz").exists() and (tmp_path / "pre" / "pr_manifold_cifar10.pt").exists()
    loaded = M.load_manifold(tmp_path / "pre" / "pr_manifold_cifar10.pt")
    assert torch.equal(loaded.features, mt.features) and torch.equal(loaded.kth, mt.kth)
    # second run: the dataset is gone, the precomputed files answer
    for fn in os.listdir(folder):
        os.remove(folder / fn)
    again = cli.main(argv, extractors={"fid": fid_net, "pr": pr_net})
    assert again == out
    lines = (tmp_path / "images" / "metrics.txt").read_text()
    assert lines.count("'folder_name': 'run1'") == 2 and "'fid'" in lines and "'pr'" in lines
    # no network given: refused
    monkeypatch.delenv("DDPM_TORCH_AMD_INCEPTION", raising=False)
    with pytest.raises(RuntimeError, match="feature network"):
        cli.main(argv + ["--metrics", "fid"])
