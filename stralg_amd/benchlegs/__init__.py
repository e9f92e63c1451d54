"""The measurement legs bench.py runs outside its timed region (host-buffer rates, the other BASELINE configurations, the
box's measured memory ceiling, counters replayed from profiles/), and the checks of what a caller of the reference-named C
API gets at BASELINE's sizes.  Measurement and test support: nothing here is on the product's data path, and nothing here
touches oracle/ (the CPU baseline leg, the one place that runs the reference / the oracle, stays in bench.py itself)."""
