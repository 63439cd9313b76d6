"""Parity and host-logic tests of the MI355X temporal-rollout path (CPU: -m "not gpu"; MI355X: -m gpu)."""
