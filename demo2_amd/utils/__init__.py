"""Host-side mirrors of the reference's utils/ for the pieces either side of the hot path (evaluator)."""
