"""Times the SMPL-X joint-chain kernel against the level-batched torch path on the device (4 and 400 frames)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soar_amd import synthetic as syn
from soar_amd.smplx_joints import JointTransformer

dev = torch.device("cuda:0")
body = syn.make_body_model(0, V=10475)
jt = JointTransformer(body.v_template, body.shapedirs, body.J_regressor, body.parents)
jt_dev = JointTransformer(body.v_template, body.shapedirs, body.J_regressor, body.parents).to(dev)
for F in (1, 4, 400):
    poses = syn.make_pose_sequence(F, 0)
    betas = torch.cat([poses["betas"].expand(F, -1), poses["expression"]], dim=1).to(dev)
    pose, transl = poses["full_pose"].to(dev), poses["transl"].to(dev)
    right = torch.linalg.inv(jt_dev(betas[:1], torch.zeros(1, 165, device=dev), None))[0]

    def t(fn, n=50):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    a = t(lambda: jt.hip(betas, pose, transl, right=right))
    b = t(lambda: torch.matmul(jt_dev(betas, pose, transl), right))
    print(f"F={F}: HIP joint chain {a:.0f} us | torch level-batched on device {b:.0f} us", flush=True)
