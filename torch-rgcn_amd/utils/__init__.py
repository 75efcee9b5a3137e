"""Host-side helpers of the experiment scripts (counterpart of the reference's top-level `utils/` package)."""
