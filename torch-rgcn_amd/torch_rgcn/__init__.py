"""torch_rgcn for AMD Instinct MI355X (gfx950): the reference's layer API on hand-written HIP kernels."""
__all__ = ["layers", "models", "utils"]
