"""The CPU side of the end-to-end parity tests lives in oracle/pipeline_ref.py (bench.py's parity leg uses it too)."""
from oracle.pipeline_ref import oracle_frames, track, oracle_events, event_signature, classify_keep          # noqa: F401
