"""Buffer on the device (rpt_buffer_*, Filter::Box + color_bytes + variance, src/buffer.rs) against the
host restatement `rpt_amd.Buffer`, which tests/test_host_api.py pins to the reference's nested loops."""
import numpy as np
import pytest

from rpt_amd import Buffer, DeviceBuffer, Filter, Renderer, RptError, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("radius,w,h", [(0, 37, 23), (1, 37, 23), (2, 16, 50), (5, 4, 3)])
def test_box_filter_bytes_and_variance_match_the_host_buffer(radius, w, h):
    rng = np.random.default_rng(radius)
    host, dev = Buffer(w, h, Filter.Box(radius)), DeviceBuffer(w, h, Filter.Box(radius))
    for k in range(4):
        batch = rng.uniform(-0.2, 1.4, (w * h, 3)) ** 2       # values below 0 / above 1 exercise the clamp
        batch[rng.integers(0, w * h, 3)] = 0.0
        host.add_samples(batch)
        dev.add_samples(batch)
    assert dev.batches == 4
    a, b = host.image(), dev.image()
    assert a.shape == b.shape == (h, w, 3)
    # same fp64 sums in the same order; pow() may differ in the last ulp, i.e. a byte may flip at an exact boundary
    assert (a != b).mean() < 1e-3 and np.abs(a.astype(int) - b.astype(int)).max() <= 1
    assert abs(dev.variance() - host.variance()) <= 1e-12 * host.variance()


def test_render_and_iterative_render_use_the_device_buffer_and_match_the_host_path():
    scene, cam, cfg = scenes.cornell()
    make = lambda: (Renderer(scene, cam).width(64).height(48).max_bounces(2).num_samples(12).filter(Filter.Box(1))  # noqa: E731
                    .seed(5))
    img = make().render()
    r = make()
    host = Buffer(64, 48, Filter.Box(1))
    r.sample(12, host)                                        # host buffer: frame downloaded as fp64
    assert np.array_equal(img, host.image())
    seen = []
    make().iterative_render(5, lambda it, buf: seen.append((it, buf.batches, buf.variance() if buf.batches > 1 else None,
                                                            buf.image())))
    assert [s[0] for s in seen] == [5, 10, 12] and [s[1] for s in seen] == [1, 2, 3]
    r2 = make()
    host2 = Buffer(64, 48, Filter.Box(1))
    for steps in (5, 5, 2):
        r2.sample(steps, host2)
    assert np.array_equal(seen[-1][3], host2.image())
    assert abs(seen[-1][2] - host2.variance()) <= 1e-9 * host2.variance()


def test_buffer_errors():
    dev = DeviceBuffer(8, 8)
    with pytest.raises(RptError):
        dev.image()                                           # "Pixel found with no samples" (buffer.rs:89)
    with pytest.raises(AssertionError):
        dev.add_samples(np.zeros((10, 3)))                    # "Invalid sample dimension" (buffer.rs:33-36)
    scene, cam, cfg = scenes.cornell()
    with pytest.raises(RptError):
        Renderer(scene, cam).width(16).height(16).sample(1, dev)
    assert np.isnan(dev.variance())
