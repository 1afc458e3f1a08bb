from .builder import (BarlowTwins, CrossCorrLossFn, cross_corr_loss_terms,  # noqa: F401
                      exchange_cross_corr)
