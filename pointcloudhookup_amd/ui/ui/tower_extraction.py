"""Drop-in for the reference's ``ui/ui/tower_extraction.py``, which is byte-identical to
``utils/tower_extraction.py`` there; here it re-exports the one implementation."""
from ...utils.tower_extraction import (  # noqa: F401
    extract_towers, extract_towers_optimized, create_obb_geometries, _save_tower_las)
