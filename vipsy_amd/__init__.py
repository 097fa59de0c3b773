"""vipsy_amd: MI355X-native ELBO-gradient engine behind the vi.py model-class surface."""
__all__ = ["engine", "_hip"]
