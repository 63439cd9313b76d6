import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sea_oracle as O
from oracle.recipe import recipe_inputs, recipe_params
from tests.test_model_gpu import build
from tests.conftest import rel_l2

L, E, H, F, B, T, ln = [int(a) if a.isdigit() else a for a in sys.argv[1:8]]
cfg = O.OracleConfig(L, E, H, 96, 8, 0, F, 2, True, ln)
p = recipe_params(cfg)
x, tgt, ib = recipe_inputs(B, T, cfg, seed=77)
_, loss_ref, grads_ref = O.loss_and_grads(x, ib, tgt, p, cfg)
m = build(cfg, "fp32").train()
eng = m.engine()
out, plan = eng.forward_train(x.cuda(), ib.cuda())
loss, dout = eng.mse_loss_and_grad(out, tgt.cuda())
eng.zero_grads()
eng.backward(plan, dout)
errs = sorted(((rel_l2(eng.grad_view(k).cpu().numpy(), g.numpy()), k) for k, g in grads_ref.items()), reverse=True)
print("cfg", sys.argv[1:], "loss", loss.item(), float(loss_ref))
for e, k in errs[:25]:
    print(f"{e:10.3e}  {k}")
print("n bad (>1e-3):", sum(e > 1e-3 for e, _ in errs), "of", len(errs))
