"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference path (see oracle.cpp, catalog_oracle.py)."""
