"""CPU oracle of the render() hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package.
Nothing under path_tracer_amd/ does; the product has no CPU path."""
