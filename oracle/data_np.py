"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the reference's input pipeline arithmetic
(/root/reference/datasetLoader.py:47-61): tf.keras image_dataset_from_directory(image_size=(S,S)) resizes with
tf.image.resize(method='bilinear') = ResizeBilinear with half-pixel centres and no antialiasing, then x / 255 and
tf.image.flip_up_down.  PARITY UNPINNED (TensorFlow is not installable here); pinned by hand-computed known
answers in tests/test_oracle.py.  Nothing under shmgan_amd/ may import this module.
"""
import numpy as np


def resize_bilinear(img, ho, wo):
    """img [H,W,C] (any real dtype) -> float32 [ho,wo,C], computed in float32 like the TF kernel."""
    img = np.asarray(img, dtype=np.float32)
    hin, win = img.shape[:2]

    def weights(n_out, n_in):
        scale = np.float32(n_in) / np.float32(n_out)
        f = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
        fl = np.floor(f)
        lo = np.maximum(fl.astype(np.int64), 0)
        hi = np.minimum(np.ceil(f).astype(np.int64), n_in - 1)
        return lo, hi, (f - fl).astype(np.float32)

    y0, y1, ly = weights(ho, hin)
    x0, x1, lx = weights(wo, win)
    lx = lx[None, :, None]
    ly = ly[:, None, None]
    top = img[y0][:, x0] + (img[y0][:, x1] - img[y0][:, x0]) * lx
    bot = img[y1][:, x0] + (img[y1][:, x1] - img[y1][:, x0]) * lx
    return (top + (bot - top) * ly).astype(np.float32)


def load_view(img_u8, size, flip_ud=True):
    """decoded uint8 RGB -> [size,size,3] float32 in [0,1] as the training loop receives it."""
    x = resize_bilinear(img_u8, size, size) * np.float32(1.0 / 255.0)
    return x[::-1].copy() if flip_ud else x
