"""rehrseg_amd -- MI355X-native 3D-convolutional hot path of zhiyuns/REHRSeg.

``rehrseg_amd.models`` mirrors the reference import surface
(models.FLAVR.FLAVR_arch.UNet_3D_3D, models.seg_model.SegModel / Distiller); put
this package directory first on sys.path to make a train_all.py-style driver
pick it up unchanged.
"""
__version__ = "0.1.0"
