"""Model assembly: the reference's `create_model` topology (contextflow/model.py:95-163) restated
without its module-global `c` (model.py:113,125 read `c.dataset`; here `config['dataset']`).

Same call signature and the same layer order, so `state_dict` keys are identical (SURVEY.md
Appendix B).  Generalist (context-free) models, and specialist models (`generalist=False`: every Conv1x1 / ActNorm /
Coupling gets its own ContextEncoder, the priors an embedding lookup — model.py:117-162) for the conv couplings with
the eye | onehot + uniform context encoders."""
from .layers import (ActNorm, Augment, ContextEncoder, Conv1x1, Coupling, Dequantization, FlowSequential,
                     GaussianMixtureDistribution, LogitTransform, MaskedCoupling, Normalization, PermuteAxes, SplitPrior, Squeeze,
                     StandardNormal, TransCoupling, UniformDistribution)

ALPHA = 1e-4
COMPONENTS = 8

# dataset -> (data_size, mixtures, num_blocks, block_size, split_prior, coupling)   model.py:173-220
PRESETS = {
    "mnist": ((1, 32, 32), 10, 2, 2, False, "conv"),
    "cifar10": ((3, 32, 32), 10, 3, 4, True, "conv"),
    "atm": ((38, 144, 1), 2, 3, 4, True, "trans"),          # model.py:189-198: 26 + 12 variables, window 144
    "smap": ((25, 8, 1), 1, 2, 4, False, "trans"),
    "msl": ((55, 8, 1), 1, 2, 4, False, "trans"),
    "smd": ((38, 8, 1), 1, 2, 4, False, "trans"),
}


def preset_config(dataset, coupling=None):
    data_size, mixtures, nb, bs, split, cpl = PRESETS[dataset]
    cfg = dict(dataset=dataset, contextflow=False, generalist=True, enc_emb="onehot", enc_type="uniform",
               num_blocks=nb, block_size=bs, actnorm=True, coupling=coupling or cpl, split_prior=split, dist="gauss")
    return cfg, data_size, mixtures


def create_model(config, data_size=(1, 1, 1), mixtures=1, contexts=(-1,)):
    generalist = config.get("generalist", True)
    contextflow = bool(config.get("contextflow", False))
    enc_emb, enc_type = config.get("enc_emb", "onehot"), config.get("enc_type", "uniform")
    contexts = list(contexts)

    def ctxnet(data_size, init="orthogonal", emb=None, typ=None):          # kwargs_context[mode] (model.py:93,117)
        if generalist:
            return None
        return ContextEncoder(contexts, emb or enc_emb, typ or enc_type, data_size, init=init)

    def prior(size):                                                       # model.py:157,162: simple lookup
        cn = ctxnet((2 * mixtures * COMPONENTS * size[0] // len(contexts),), init="zeros", emb="embed", typ="eyesample")
        return GaussianMixtureDistribution(size=size, mixtures=mixtures, components=COMPONENTS, context_net=cn,
                                           contextflow=contextflow)
    if not generalist and config["coupling"] not in ("conv", "trans"):
        raise NotImplementedError("specialist models: --coupling conv | trans are built (SURVEY.md 8(f) rank 2)")
    if config.get("dist", "gauss") != "gauss":
        raise NotImplementedError("only the Gaussian-mixture prior is implemented")
    dataset = config["dataset"]
    layers = []
    if dataset in ("mnist", "cifar10"):
        layers += [Dequantization(UniformDistribution(size=data_size)),
                   Normalization(translation=0.0, scale=256.0),
                   Normalization(translation=ALPHA, scale=1 / (1 - 2 * ALPHA)),
                   LogitTransform()]
    ts = dataset in ("atm", "msl", "smd", "smap")
    patch, krn, pad = ((2, 1), (3, 1), (1, 0)) if ts else ((2, 2), (3, 3), (1, 1))
    sz = tuple(data_size)
    for blk in range(config["num_blocks"]):
        if sz[0] % 2:
            layers.append(Augment(StandardNormal((1, sz[1], sz[2])), 1))
            sz = (sz[0] + 1, sz[1], sz[2])
        if dataset not in ("msl", "smd", "smap"):
            layers.append(Squeeze(patch_size=patch))
            sz = (sz[0] * patch[0] * patch[1], sz[1] // patch[0], sz[2] // patch[1])
        for _ in range(config["block_size"]):
            layers.append(Conv1x1(sz, context_net=ctxnet((sz[0],), init="zeros"), contextflow=contextflow))
            if config["actnorm"]:
                layers.append(ActNorm(sz, context_net=ctxnet((2 * sz[0],)), contextflow=contextflow))
            if config["coupling"] == "trans" and sz[1] % patch[0] == 0 and sz[2] % patch[1] == 0:
                layers.append(TransCoupling(sz, patch, context_net=ctxnet((sz[0],)), contextflow=contextflow))
            elif config["coupling"] == "conv":
                layers.append(Coupling(sz[0], kernel_size=krn, padding=pad, context_net=ctxnet((sz[0],)),
                                       contextflow=contextflow))
            elif config["coupling"] == "maf":
                layers.append(MaskedCoupling(sz[0], kernel_size=krn, padding=pad, context_net=ctxnet((sz[0],)),
                                             contextflow=contextflow))
            if dataset == "atm":                          # model.py:149-151: swap channel and time axes
                layers.append(PermuteAxes((0, 2, 1, 3)))
                sz = (sz[1], sz[0], sz[2])
        if config["split_prior"] and blk < config["num_blocks"] - 1:
            sz = (sz[0] // 2, sz[1], sz[2])
            layers.append(SplitPrior(prior(sz)))
    return FlowSequential(prior(sz), *layers)
