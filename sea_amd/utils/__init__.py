"""Rollout helpers, scalers and dataset windows around the HIP engine."""
