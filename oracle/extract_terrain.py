"""terrain_image_100.png of the reference -> tests/golden/terrain_100.npy (uint8 [100,100], image row order).
Runs only in the build container (PIL + the reference checkout); the array is data."""
import os
import numpy as np
from PIL import Image

REF = "/root/reference/terrain_image_100.png"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "terrain_100.npy")
a = np.array(Image.open(REF))
assert a.shape == (100, 100, 3) and np.array_equal(a[:, :, 0], a[:, :, 1]) and np.array_equal(a[:, :, 1], a[:, :, 2])
np.save(OUT, a[:, :, 0].astype(np.uint8))
print(OUT, a[:, :, 0].min(), a[:, :, 0].max())
