"""Image helpers replacing scipy.misc.toimage / imread and ImageMagick `convert` used by the reference's
example scripts (none of them exists in this environment)."""
import numpy as np
from PIL import Image


def to_uint8(image, cmin=0.0, cmax=1.0):
    image = np.asarray(image, np.float32)
    return (np.clip((image - cmin) / (cmax - cmin), 0, 1) * 255).astype(np.uint8)


def save_image(image, filename):
    Image.fromarray(to_uint8(image)).save(filename)


def read_image(filename):
    return np.asarray(Image.open(filename)).astype(np.float32) / 255.0


def make_gif(frames, filename, duration=40):
    ims = [Image.fromarray(to_uint8(f)) for f in frames]
    ims[0].save(filename, save_all=True, append_images=ims[1:], duration=duration, loop=0)
