import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DVS_TR_DEBUG"] = "32"
import numpy as np, torch
import divshot_amd as dv
from divshot_amd.raster import Rasterizer, params_to_device
from bench import WORKLOADS
n, W, H, deg, soff = WORKLOADS["C3"]
spec = dv.make_spec(n, W, H, sh_degree=deg, n_cams=8, scale_log_offset=soff)
dev = torch.device("cuda", 0)
P = params_to_device(dv.synth_splats(spec), dev)
cam = dv.synth_camera(spec, 0)
tgt = torch.from_numpy(dv.synth_target(spec, 0)).to(dev)
r = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
r.set_backward_variant("tr64")
img = r.forward(P, cam, sh_degree=deg, absgrad=True)
dL = ((img - tgt) / (W * H)).contiguous()
out = (C.c_ulonglong * 8)()
dv.lib.dvs_tr_debug_counters(out, 1)
r.backward_composite(dL); r.backward_project(); torch.cuda.synchronize()
dv.lib.dvs_tr_debug_counters(out, 1)
print(json.dumps({"lib": os.environ.get("DVS_RASTER_LIB", "default"), "rounds": out[0], "rounds_with_shared_row": out[1], "steps": out[2], "skipped_steps": out[3], "batch_waves": out[4], "sum_list_len": out[5], "productive_pairs": out[6], "contributing_pixel_pairs": out[7], "T": int(r.get_num_rendered())}))
