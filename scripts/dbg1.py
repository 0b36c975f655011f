import sys, importlib; sys.path.insert(0, '.')
import numpy as np
pkg = importlib.import_module('mpc-sensorlessao_amd')
from tests.util import handle_from_model, oracle_batch, rel_err
def run(model, data, nw=1, k=1e-2, tag=''):
    h = handle_from_model(pkg, model)
    z, info = h.solve(data["x0"], data.get("x0_pre"), data.get("w"), nu0=data.get("nu0"), n_newton=nw, k=k, return_info=True, check=False)
    zo, nuo, ito, sto, steps = oracle_batch(model, data, nw, k)
    T, s, m = model['T'], model['n']+model['m'], model['m']
    e = [rel_err(z[p], zo[p]) for p in range(z.shape[0])]
    Z = z.reshape(-1, T, s); Zo = zo.reshape(-1, T, s)
    print(tag, 'max rel', max(e), 'u err', rel_err(Z[:, :, :m], Zo[:, :, :m]), 'x err', rel_err(Z[:, :, m:], Zo[:, :, m:]), 'nu err', rel_err(info['nu'], nuo), 'iters', info['iters'][:4], ito[:4], flush=True)
    h.close()
S = pkg.synthetic
for (n, m, T) in [(27, 144, 2), (27, 144, 1), (16, 144, 2), (27, 20, 2), (8, 5, 2), (27, 144, 3)]:
    model = S.make_model(n, m, T); data = S.make_replay_batch(model, r=1, steps=4)
    run(model, data, tag=f'ao n{n} m{m} T{T}')
    model2 = dict(model); model2['Q'] = np.eye(n); model2['Qf'] = np.eye(n)
    run(model2, data, tag=f'ao Q=I n{n} m{m} T{T}')
model, data = S.make_test_problem(8, 5, 10, seed=3, batch=4)
model['Q'] = 1.5e4*np.eye(8); model['Qf'] = 1.5e4*np.eye(8)
run(model, data, tag='demo Q=1.5e4')
