"""Signal transforms of the autoencoder path (reference ``behavenet/data/transforms.py``).

Only what the conv-AE data path uses: :class:`MakeOneHot2D` turns the ``labels_sc`` pixel
coordinates into the one-hot 2-d maps a conditional encoder takes as extra input channels
(reference transforms.py:186-245, selected at data/utils.py:96-99).
"""

import numpy as np

__all__ = ['Transform', 'Compose', 'MakeOneHot2D']


class Transform(object):
    """Callable on a (time, ...) numpy array."""

    def __call__(self, sample):
        raise NotImplementedError

    def __repr__(self):
        return '%s()' % type(self).__name__


class Compose(Transform):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample

    def __repr__(self):
        return 'Compose(%s)' % ', '.join(repr(t) for t in self.transforms)


class MakeOneHot2D(Transform):
    """(time, 2 * n_labels) coordinates -> (time, n_labels, y_pixels, x_pixels) one-hot maps.

    The first half of the columns are x values, the second half y values; each is rounded and
    clipped into the image, NaN counts as 0.  Label n of frame t lights pixel (y, x) of map n:
    with 128 x 128 maps, ``[64, 34, 56, 102]`` sets ``out[0, 56, 64]`` and ``out[1, 102, 34]``.
    """

    def __init__(self, y_pixels, x_pixels):
        self.y_pixels = y_pixels
        self.x_pixels = x_pixels

    def _pixel(self, vals, size):
        vals = np.where(np.isnan(vals), -1.0, vals)
        return np.round(np.clip(vals, 0, size - 1)).astype(np.int64)

    def __call__(self, sample):
        n_time, n_cols = sample.shape
        n_labels = n_cols // 2
        xs = self._pixel(np.asarray(sample[:, :n_labels], dtype=np.float64), self.x_pixels)
        ys = self._pixel(np.asarray(sample[:, n_labels:2 * n_labels], dtype=np.float64),
                         self.y_pixels)
        out = np.zeros((n_time, n_labels, self.y_pixels, self.x_pixels))
        t = np.arange(n_time)
        for n in range(n_labels):
            out[t, n, ys[:, n], xs[:, n]] = 1
        return out

    def __repr__(self):
        return 'MakeOneHot2D(y_pixels=%i, x_pixels=%i)' % (self.y_pixels, self.x_pixels)
