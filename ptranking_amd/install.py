"""Drop the fused rankers into an installed wildltr/ptranking so that its unchanged pipeline driver picks them up.

`LTREvaluator.load_ranker` instantiates rankers through `globals()[model_id]` of the module
ptranking/ltr_adhoc/eval/ltr.py (:156-178), so rebinding the six names in that module is all a drop-in needs:

    import ptranking_amd
    ptranking_amd.install()          # RankNet, LambdaRank, LambdaLoss, ApproxNDCG, ListNet, ListMLE -> fused HIP versions
    LTREvaluator(cuda=0).run(model_id='LambdaRank', ...)   # the reference's own driver, data layer, config, tapes

The installed classes derive from the reference's own AdhocNeuralRanker (ptranking/base/adhoc_ranker.py:7), i.e. the
scorers (pointsf AND listsf), optimiser config, save/load stay the reference's code; only `custom_loss_function`,
the train loop's host syncs and the Evaluator metric methods are replaced.  `<Model>Parameter` classes are left alone.
"""
import importlib

from .rankers import EXTRA_RANKER_NAMES, RANKER_NAMES, make_ranker_classes

_saved = {}


def install(names=RANKER_NAMES, ltr_module="ptranking.ltr_adhoc.eval.ltr", extras=False):
    """Rebind `names` inside the reference's ltr module; returns {name: installed class}.  extras=True adds the rankers SURVEY.md 2 marks
    out of scope (EXTRA_RANKER_NAMES: DASALC, MDPRank), which the default drop-in leaves alone."""
    if extras:
        names = tuple(names) + tuple(n for n in EXTRA_RANKER_NAMES if n not in names)
    mod = importlib.import_module(ltr_module)
    base = importlib.import_module("ptranking.base.adhoc_ranker").AdhocNeuralRanker
    classes = make_ranker_classes(base)
    done = {}
    for n in names:
        if n not in classes:
            raise KeyError(f"{n} is not one of {RANKER_NAMES + EXTRA_RANKER_NAMES}")
        _saved.setdefault((ltr_module, n), getattr(mod, n, None))
        setattr(mod, n, classes[n])
        done[n] = classes[n]
    return done


def uninstall(ltr_module="ptranking.ltr_adhoc.eval.ltr"):
    """Restore the reference's own classes."""
    mod = importlib.import_module(ltr_module)
    for (m, n), cls in list(_saved.items()):
        if m != ltr_module:
            continue
        if cls is not None:
            setattr(mod, n, cls)
        elif hasattr(mod, n):          # the name did not exist before install() (e.g. DASALC is never imported by ltr.py)
            delattr(mod, n)
        del _saved[(m, n)]
