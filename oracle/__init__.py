"""CPU oracle: a plain-PyTorch fp32 restatement of the reference's forward pass.

TEST INFRASTRUCTURE ONLY.  Nothing under ``resselt_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and only as the checker
or the reported CPU baseline -- never as the thing measured or shipped.

Every function cites the reference lines (under /root/reference) it restates.  The oracle is pinned
against the real reference: ``tools/gen_golden.py`` imports rewaifu/resselt in the build container,
runs its modules on seeded inputs with deterministic weights and commits the input/output vectors
under ``tests/golden/``; ``tests/test_oracle_golden.py`` replays them through this package.
"""
