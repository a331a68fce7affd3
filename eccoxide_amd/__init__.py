"""eccoxide_amd -- MI355X-native batched scalar multiplication behind eccoxide's
CurveGroup / Point / Scalar surface (hot path only)."""
from .engine import (  # noqa: F401
    BLS12_381_G1, CURVE_IDS, CURVE_NAMES, ED25519, FLAG_FINITE, FLAG_INFINITY, FLAG_REJECTED, P256R1, P384R1,
    P521R1, EccxError, Engine, curve_id, field_bytes, scalar_bytes,
)
