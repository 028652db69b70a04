"""`rehrseg_amd.models`; also importable as the top-level package `models` the reference's train_all.py:20-31
names, when the directory rehrseg_amd/ is put on sys.path."""
if __name__ == "models":  # found as a top-level package: hand over to rehrseg_amd.models (one set of module objects)
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    if _root not in _sys.path:
        _sys.path.append(_root)
    from rehrseg_amd import _dropin
    _dropin.alias("models")
