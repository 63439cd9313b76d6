"""Training entry points of the temporal model on the HIP engine (forward + backward + AdamW plans)."""
