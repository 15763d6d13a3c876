"""bench.py's legs as modules (timing loop, decoder probes + roofline, side configurations, launch plumbing)."""
