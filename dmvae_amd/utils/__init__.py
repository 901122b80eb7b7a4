"""Host-side mirror of the reference's utils used on the hot path (utils/lpips.py, utils/dist.py)."""
