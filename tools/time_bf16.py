import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minizero_amd as mz
for key in ("c2", "c3"):
    d = mz.DESCS[key]()
    net = mz.Net(d, mz.generate_weights(d, 0))
    B = 256 if key == "c2" else 1024
    for prec in ("f32", "bf16x3"):
        net.set_precision(prec)
        ms_fwd, ms_tower, fl = net.time_forward(B, 200)
        print(key, prec, "B", B, "tower us %.1f" % (ms_tower * 1e3), "forward us %.1f" % (ms_fwd * 1e3), "f32-equivalent TFLOP/s %.1f" % (fl / (ms_tower * 1e-3) / 1e12), flush=True)
