"""Data formats either side of the hot path: Middlebury .flo files, the EPE metric, the reference's crop rule."""
