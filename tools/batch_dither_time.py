import sys, time, numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
from oracle import binding as ob
w = h = 512; K = 64; count = 24
n = w * h
imgs = [np.asfortranarray(ob.image(n, 300 + i).reshape(3, n).T) for i in range(count)]
t = time.time(); one = [p.quantize(w, h, im, K, dither=True, tile_size=0, kmeans_niter=2, kmeans_max_samples=4096) for im in imgs[:4]]; t1 = (time.time() - t) / 4
t = time.time(); res = p.quantize_batch(w, h, imgs, K, dither=True, tile_size=0, kmeans_niter=2, kmeans_max_samples=4096); tb = time.time() - t
print("single %.1f ms/image; batch of %d: %.1f ms/image (first call: engines are created)" % (t1 * 1e3, count, tb * 1e3 / count))
t = time.time(); res = p.quantize_batch(w, h, imgs, K, dither=True, tile_size=0, kmeans_niter=2, kmeans_max_samples=4096); tb = time.time() - t
print("second batch call: %.1f ms/image" % (tb * 1e3 / count))
print("batch == single:", all(np.array_equal(res[i][2], one[i][2]) and np.array_equal(res[i][1], one[i][1]) for i in range(4)))
