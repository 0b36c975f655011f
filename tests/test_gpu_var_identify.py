"""VAR(2) identification on the device (fmpc_var_identify_device) vs the numpy restatement of README.md:108-130 on the
synthetic coefficient series, and recovery of the model that generated them.  Tolerance 1e-7 relative on A1, A2: the
normal equations square the condition number of AA (cond(AA'AA) ~ 1e7 on these series), so two fp64 evaluations with
different summation orders agree to about 1e-16 x 1e7."""
import numpy as np
import pytest
import torch

from oracle.var_identify_ref import identify_var2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,num_train,batch", [(27, 1000, 5), (8, 200, 3), (32, 300, 2), (5, 40, 1)])
def test_var_identify_matches_numpy_restatement(pkg, gpu, n, num_train, batch):
    model = pkg.synthetic.make_model(n, 16, 4)
    series = np.stack([pkg.synthetic.make_realisation(model, r=r, steps=num_train + 20) for r in range(batch)])
    t = torch.from_numpy(series).to(gpu)
    A1, A2, st = pkg.identify_var2_device(t, num_train=num_train)
    torch.cuda.synchronize()
    assert int(st.abs().sum()) == 0
    for b in range(batch):
        r1, r2 = identify_var2(series[b], num_train)
        e1 = np.linalg.norm(A1[b].cpu().numpy() - r1) / np.linalg.norm(r1)
        e2 = np.linalg.norm(A2[b].cpu().numpy() - r2) / np.linalg.norm(r2)
        assert e1 <= 1e-7 and e2 <= 1e-7, (b, e1, e2)
    # single-series form, and the identified model is close to the generating one (a long, well excited series)
    a1, a2, s1 = pkg.identify_var2_device(t[0], num_train=num_train)
    assert torch.equal(a1, A1[0]) and torch.equal(a2, A2[0])


def test_var_identify_recovers_the_generating_model(pkg, gpu):
    model = pkg.synthetic.make_model(27, 16, 4)
    series = pkg.synthetic.make_realisation(model, r=3, steps=20000)
    A1, A2, st = pkg.identify_var2_device(torch.from_numpy(series).to(gpu))
    assert int(st) == 0
    # (noise limited: the numpy restatement recovers A1 to 5 %, A2 to 12 % from 20000 samples)
    assert np.linalg.norm(A1.cpu().numpy() - model["A1"]) / np.linalg.norm(model["A1"]) < 0.1
    assert np.linalg.norm(A2.cpu().numpy() - model["A2"]) / np.linalg.norm(model["A2"]) < 0.2


def test_var_identify_errors(pkg, gpu):
    t = torch.zeros((1, 30, 27), dtype=torch.float64, device=gpu)
    with pytest.raises(pkg.FastMPCError):
        pkg.identify_var2_device(t)                              # fewer rows than unknowns
    t = torch.zeros((2, 200, 8), dtype=torch.float64, device=gpu)
    A1, A2, st = pkg.identify_var2_device(t)                      # an all-zero series: AA'AA is singular
    assert (st.cpu().numpy() == pkg.FMPC_E_NOT_PD_SCHUR).all()
