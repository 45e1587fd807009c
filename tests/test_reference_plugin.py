"""Plug-in plumbing against the REAL reference (build container only: needs /root/reference; the GPU
box skips these).  No GPU arithmetic runs here -- what is checked is that our objects fit the
reference's extension points:

  * elfi.Distance / elfi.Discrepancy accept HipDistance / HipDiscrepancy as operations, the model
    compiles, and the compiled + loaded net pickles (tests/functional/test_serialization.py);
  * elfi.BOLFI accepts HipGPRegression / HipLCBSC as target_model / acquisition_method;
  * the multi-GPU client (elfi_amd/gpu_client.py) drives a real Rejection run through worker
    processes and reproduces the native client's sample bit for bit (the operations executed in
    the workers are the reference's own CPU ones, so this runs without a GPU).
"""
import os
import pickle
import sys

import numpy as np
import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')
sys.path.insert(0, ORACLE)
import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason='/root/reference not present')


@pytest.fixture(scope='module')
def elfi():
    return ref_shim.install()


def _ma2(elfi, seed_obs=4):
    from elfi.examples import ma2
    return ma2.get_model(seed_obs=seed_obs)


def test_hip_operations_fit_the_node_api_and_pickle(elfi):
    import elfi_amd
    m = _ma2(elfi)
    d1 = elfi.Distance(elfi_amd.HipDistance('euclidean'), m['S1'], m['S2'], name='d_hip')
    d2 = elfi.Discrepancy(elfi_amd.HipDiscrepancy('minkowski', p=3), m['S1'], m['S2'], name='d_hip2')
    s3 = elfi.Summary(elfi_amd.autocov, m['MA2'], 2, name='S3')
    assert d1.state['attr_dict']['_operation'] is not None and d2.state['attr_dict']['_operation'] is not None
    assert s3.parents[0].name == 'MA2'
    # model save/copy path: operations must survive pickle (elfi_model.py:401-438)
    blob = pickle.dumps(m)
    m2 = pickle.loads(blob)
    assert set(m2.nodes) >= {'d_hip', 'd_hip2', 'S3'}
    # compiled + loaded net pickles with our operations inside (tests/functional/test_serialization.py:28-43)
    from elfi.client import ClientBase
    from elfi.model.elfi_model import ComputationContext
    compiled = ClientBase.compile(m.source_net, ['d_hip', 'd_hip2', 'S3'])
    loaded = ClientBase.load_data(compiled, ComputationContext(), 0)
    loaded2 = pickle.loads(pickle.dumps(loaded))
    op = loaded2.nodes['d_hip2']['operation'] if 'operation' in loaded2.nodes['d_hip2'] else None
    assert op is not None


def test_bolfi_accepts_the_hip_surrogate_and_acquisition(elfi):
    import elfi_amd
    m = _ma2(elfi)
    log_d = elfi.Operation(np.log, m['d'], name='log_d')
    bounds = {'t1': (-2, 2), 't2': (-1, 1)}
    gp = elfi_amd.HipGPRegression(['t1', 't2'], bounds=bounds)
    acq = elfi_amd.HipLCBSC(gp, noise_var=0.1, exploration_rate=10, seed=1)
    bolfi = elfi.BOLFI(log_d, batch_size=1, initial_evidence=5, update_interval=10, bounds=bounds,
                       target_model=gp, acquisition_method=acq, seed=1)
    assert bolfi.target_model is gp and bolfi.acquisition_method is acq
    assert bolfi.target_model.parameter_names == ['t1', 't2']
    # the duck-type the reference's BO code reads (bolfi.py:86-137, acquisition.py:37-41)
    for attr in ('input_dim', 'bounds', 'n_evidence', 'predict', 'predict_mean', 'predictive_gradients',
                 'predictive_gradient_mean', 'update', 'optimize', 'copy', 'is_sampling'):
        assert hasattr(gp, attr), attr
    assert gp.n_evidence == 0 and gp.bounds == [(-2, 2), (-1, 1)]


def test_surrogate_is_an_instance_of_the_reference_class(elfi):
    """bolfire.py:329-331 insists on `isinstance(target_model, GPyRegression)`: with the reference imported, objects of
    HipGPRegression (and of user subclasses) are instances of both classes; our methods come first, pickling and
    copying go through the importable class."""
    import copy
    import elfi_amd
    from elfi.methods.bo.gpy_regression import GPyRegression
    gp = elfi_amd.HipGPRegression(['t1', 't2'], bounds={'t1': (-2, 2), 't2': (-1, 1)})
    assert isinstance(gp, GPyRegression) and isinstance(gp, elfi_amd.HipGPRegression)
    assert type(gp).__name__ == 'HipGPRegression' and type(gp).__mro__[1] is elfi_amd.HipGPRegression
    assert type(gp).predict is elfi_amd.HipGPRegression.predict and type(gp).update is elfi_amd.HipGPRegression.update
    gp2 = pickle.loads(pickle.dumps(gp))
    assert isinstance(gp2, GPyRegression) and gp2.bounds == gp.bounds and gp2.parameter_names == ['t1', 't2']
    assert isinstance(copy.copy(gp), GPyRegression) and isinstance(gp.copy(), GPyRegression)

    class Mine(elfi_amd.HipGPRegression):
        pass
    assert isinstance(Mine(['a'], bounds={'a': (0, 1)}), GPyRegression)
    assert type(elfi_amd.HipGPRegression(['a'], bounds={'a': (0, 1)})) is type(gp)      # one derived class, cached


@pytest.mark.timeout(180)
def test_gpu_client_reproduces_the_native_client(elfi):
    import elfi.clients.native as native
    m = _ma2(elfi)
    native.set_as_default()
    ref = elfi.Rejection(m['d'], batch_size=500, seed=123).sample(50, n_sim=3000)
    import elfi_amd.gpu_client as gc
    # spawned workers must be able to import ref_shim (to unpickle the initializer) and elfi_amd
    root = os.path.dirname(ORACLE)
    os.environ['PYTHONPATH'] = os.pathsep.join([ORACLE, root, os.environ.get('PYTHONPATH', '')])
    client = gc.Client(num_gpus=2, worker_setup=ref_shim.install)
    try:
        elfi.set_client(client)
        got = elfi.Rejection(m['d'], batch_size=500, seed=123).sample(50, n_sim=3000)
    finally:
        client.reset()
        native.set_as_default()
    assert got.n_sim == ref.n_sim == 3000
    assert got.threshold == ref.threshold
    for k in ('t1', 't2'):
        assert np.array_equal(got.samples[k], ref.samples[k])
    assert client.num_gpus == 2 and client.num_cores == 4     # two batches in flight per GPU


def test_hip_bolfi_is_the_reference_class_with_device_defaults(elfi):
    """elfi_amd.HipBOLFI: a subclass of the running ELFI's BOLFI whose defaults are the device objects (no GPU is touched
    before the first evidence arrives); objects the caller passes are kept, and a surrogate that is not a device model
    takes the reference's own extract_posterior / sample."""
    import elfi_amd
    from oracle_gp_model import OracleGPRegression
    m = _ma2(elfi)
    log_d = elfi.Operation(np.log, m['d'], name='log_d')
    bounds = {'t1': (-2, 2), 't2': (-1, 1)}
    b = elfi_amd.HipBOLFI(log_d, batch_size=1, initial_evidence=20, update_interval=10, bounds=bounds,
                          acq_noise_var=0.1, seed=3)
    assert isinstance(b, elfi.BOLFI) and type(b).__name__ == 'HipBOLFI'
    assert elfi_amd.hip_bolfi_class() is type(b)
    assert isinstance(b.target_model, elfi_amd.HipGPRegression) and b.target_model.bounds == [(-2, 2), (-1, 1)]
    acq = b.acquisition_method
    assert isinstance(acq, elfi_amd.HipLCBSC) and acq.model is b.target_model and acq.seed == b.seed
    assert acq.exploration_rate == 10 and acq.prior is not None
    with pytest.raises(ValueError, match='not fitted'):
        b.extract_posterior()
    # a caller's own surrogate: the reference's path, unchanged
    pre = {'t1': np.linspace(-1, 1, 30), 't2': np.linspace(-0.5, 0.5, 30)}
    pre['log_d'] = np.log(0.1 + (pre['t1'] - 0.6) ** 2 + (pre['t2'] - 0.2) ** 2)
    tm = OracleGPRegression(['t1', 't2'], bounds=bounds)
    b2 = elfi_amd.HipBOLFI(log_d, batch_size=1, initial_evidence=pre, bounds=bounds, target_model=tm, seed=3)
    assert b2.target_model is tm
    from elfi.methods.bo.acquisition import LCBSC
    from elfi.methods.posteriors import BolfiPosterior
    assert type(b2.acquisition_method) is LCBSC
    assert type(b2.extract_posterior(-1.0)) is BolfiPosterior
    res = b2.sample(20, n_chains=2, threshold=-1.0, n_evidence=30)
    assert res.chains.shape == (2, 20, 2)
