"""MI355X-native coloured sparse-Jacobian finite differencing (FiniteDiff.jl hot path).

Host-side mirror of the reference's interface for this one path
(``JacobianCache`` + ``finite_difference_jacobian!``; reference: src/jacobians.jl:1-128,
446-471, 504-653) on top of the C-ABI library ``libfdjac`` (include/fdjac.h), whose kernels
are hand-written HIP for gfx950.  There is no CPU fallback: every compute entry point
raises if the HIP library or a GPU is missing.
"""
from . import patterns  # noqa: F401  (numpy-only helpers; safe without a GPU)
from . import lib  # noqa: F401  (ctypes binding; loads libfdjac.so on first use)
from .api import (BandedBlockBandedMatrix, BandedMatrix, BlockBandedMatrix, Bidiagonal, BuiltinF, Comm, Context, DevicePatternCSC, Diagonal, JacobianCache, JitF, JitTerms, BitcodeF, TrackedCSC, TrackedVector, P2P, Plan, SymTridiagonal,  # noqa: F401
                  SparseMatrixCSC, TorchF, Tridiagonal, TridiagSolver, BandedSolver, BlockTridiagSolver, JVPCache, default_relstep, finite_difference_jacobian,
                  finite_difference_jacobian_b,
                  finite_difference_jvp_b, make_plan, make_plan_csc_device, matrix_colors)

__all__ = ["patterns", "lib", "BandedBlockBandedMatrix", "BandedMatrix", "BlockBandedMatrix", "Bidiagonal", "BuiltinF", "Comm", "Context", "DevicePatternCSC", "Diagonal", "JacobianCache", "JitF", "JitTerms", "BitcodeF", "TrackedCSC", "TrackedVector", "P2P", "Plan", "SymTridiagonal",
           "SparseMatrixCSC", "TorchF", "Tridiagonal", "TridiagSolver", "BandedSolver", "BlockTridiagSolver", "JVPCache", "default_relstep", "finite_difference_jacobian", "finite_difference_jacobian_b",
           "finite_difference_jvp_b", "make_plan", "make_plan_csc_device", "matrix_colors"]
