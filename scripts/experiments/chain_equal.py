import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import renderer_amd as ra
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
scene = ra.HostScene.load_config(os.path.join(ROOT, "Data", sys.argv[2]))
cam = scene.camera(); dev = ra.Device(device=0); dev.upload(scene.describe())
acc = torch.zeros(cam.width * cam.height * 3, dtype=torch.float32, device="cuda")
st = dev.render_pt_dev(cam, acc.data_ptr(), 96, seed=7, spp_per_launch=16)
torch.cuda.synchronize()
np.save(sys.argv[1], acc.cpu().numpy())
print(sys.argv[1], "rays", st.n_rays, "mean", float(acc.mean()) / 96)
