"""CPU oracles for the heliostat render path — TEST INFRASTRUCTURE ONLY.

``torch_oracle``  materialising PyTorch restatement (bit-exact with the reference,
                  forward and autograd); also bench.py's cpu_baseline ("port").
``c_oracle``      ctypes loader for helio_oracle.c, the scalar C restatement.

Nothing under ``doodle_amd/`` imports this package.
"""
