import sys, time, numpy as np
sys.path.insert(0, ".")
import patolette_amd as p
rng=np.random.default_rng(0)
img8=rng.integers(0,256,size=(4096,4096,3),dtype=np.uint8)
colors=img8.reshape(-1,3).astype(np.float64)/255
fcol=np.asfortranarray(colors)
for it in range(3):
    r = r2 = r8 = None            # releasing a 134 MB map costs ~5 ms: keep it out of the timings
    t=time.time(); r=p.quantize(4096,4096,colors,256,dither=False,tile_size=0); a=time.time()-t
    t=time.time(); r2=p.quantize(4096,4096,fcol,256,dither=False,tile_size=0); b=time.time()-t
    t=time.time(); r8=p.quantize_u8(img8,256,dither=False,tile_size=0); c=time.time()-t
    print("h2h C-order %.1f ms  F-order %.1f ms  u8 %.1f ms"%(a*1e3,b*1e3,c*1e3), np.array_equal(r[2], r2[2]), np.array_equal(r[2], r8[2].reshape(-1)))
