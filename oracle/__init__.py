"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements (plain PyTorch fp32 on CPU) of the reference's rollout hot path, plus the
tooling that imports the real reference (from /root/reference, build container only) to pin
those restatements and to generate the golden fixtures under tests/golden/.

Nothing under dlwp_benchmark_amd/ imports this package.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may -- and there only as the checker / the reported CPU
baseline, never as the thing measured or shipped.
"""
