"""GPU: the Flask contract and the batch sharder on the real HIP backend (through the C-ABI)."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


def test_contract_replay_on_gpu_backend(dev, golden):
    import ipx.app as app_mod
    from test_host_contract import replay_contract
    client = app_mod.create_app(backend=app_mod.GpuBackend("cuda:0")).test_client()
    assert replay_contract(client, golden) >= 70


def test_opencv_backed_modes_on_gpu_backend(dev):
    import ipx.app as app_mod
    import test_host_contract as T
    client = app_mod.create_app(backend=app_mod.GpuBackend("cuda:0")).test_client()
    T.test_opencv_backed_modes_contract(client)
    T.test_extension_modes(client)
    T.test_batch_endpoint(client)
    T.test_median_modes_contract(client)


def test_batch_sharder_two_workers_one_gpu(dev):
    from ipx.sharder import BatchSharder
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (64 + 8 * i, 200, 3), dtype=np.uint8) for i in range(9)]
    sh = BatchSharder(devices=["cuda:0", "cuda:0"])                 # two host threads / streams on the one GPU here
    outs = sh.map(imgs, lambda ops, t: (ops.laplacian_sharpen(t), ops.bgr2gray_equalize(t)[1]))
    for i, (s, e) in enumerate(outs):
        np.testing.assert_array_equal(s, O.laplacian_sharpen_u8(imgs[i]))
        np.testing.assert_array_equal(e, O.equalize_hist_u8(O.bgr2gray_u8(imgs[i])))


def test_concurrent_requests_threads(dev):
    """Flask worker threads call the library concurrently: per-thread streams, no shared state."""
    import threading
    import ipx.app as app_mod
    be = app_mod.GpuBackend("cuda:0")
    rng = np.random.default_rng(6)
    imgs = [rng.integers(0, 256, (300, 400, 3), dtype=np.uint8) for _ in range(8)]
    res = [None] * 8

    def work(i):
        for _ in range(5):
            res[i] = be.run(lambda ops, t: ops.gaussian_blur(t, 5), imgs[i])

    ths = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for i in range(8):
        np.testing.assert_array_equal(res[i], O.gaussian_blur_u8(imgs[i], 5))
