"""Host-side helpers of the callers either side of the hot path: image geometry (``util/cv.py`` of
the reference) and the file formats its scripts read and write (``util/io.py``)."""
