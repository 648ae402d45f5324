"""Host-side mirror of the reference's ``models`` package (models/__init__.py:3-15)."""


def get_model_class(opt):
    if opt.model == 'pix2pix':
        from .Pix2Pix import Pix2PixModel
        return Pix2PixModel
    if opt.model == 'cyclegan':
        from .CycleGAN import MobileCycleGANModel
        return MobileCycleGANModel
    if opt.model == 'sagan':
        from .SAGAN import SAGANModel
        return SAGANModel
    if opt.model == 'srgan':
        from .SRGAN import SRGAN
        return SRGAN
    raise NotImplementedError('%s not implemented' % opt.model)
