"""CPU restatement of the reference temporal-rollout path: test infrastructure only (tests/, __graft_entry__.smoke(), bench.py cpu_baseline)."""
