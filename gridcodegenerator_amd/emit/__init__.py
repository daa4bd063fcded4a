"""Generation-time machinery: robot spec, tracer, traced algorithms, header emission."""
