"""TEST-ONLY stand-in for the `ogb` package so that the reference's scripts can be executed unmodified at toy
scale (tests/test_reference_scripts.py).  Datasets are tiny seeded synthetic graphs with OGB's object layout."""
