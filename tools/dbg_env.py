import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import golden
from doodle_amd import HelioField
g = golden("g6_env_train_n50_b25_r64")
DEV="cuda"
suns = torch.from_numpy(g["suns"])
f = HelioField(torch.from_numpy(g["helios"]).to(DEV), torch.tensor([0.,-5.,0.],device=DEV), (15.,15.), torch.tensor([0.,1.,0.],device=DEV),
               sigma_scale=0.01, error_scale_mrad=0.0, resolution=64, max_batch_size=25, device=DEV)
print("errs", f.batch_error_angles_mrad.abs().max().item(), f._trig_of(f.batch_error_angles_mrad)[0,0])
f.initial_action_noise = 0.0
f.init_actions(suns.to(DEV))
ideal = f.calculate_ideal_normals(suns.to(DEV))
print("init vs ideal", (f.initial_action.view(25,50,3)-ideal).abs().max().item())
img,_ = f.render(suns.to(DEV), f.initial_action, ideal)
img2,_ = f.render(suns.to(DEV), ideal.flatten(1), ideal)
print("img peak", img.max().item(), img2.max().item(), (img-img2).abs().max().item())
i0 = img[0].cpu().numpy(); print("argmax", np.unravel_index(i0.argmax(), i0.shape), "hot", (i0>0.5*i0.max()).sum())
# compare with golden step target? reset_img is noisy. Use distance map: region where map==0
dm = g["distance_maps"][0]; print("golden hot", (dm==0).sum(), np.argwhere(dm==0)[:5])
print(np.argwhere(i0>0.5*i0.max())[:5])
