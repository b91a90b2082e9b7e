"""flexynesis_amd.adapters without a GPU: the adapter functions carry the reference's signatures and read only attributes the
reference's classes really set (checked against the reference's source where it exists -- the build container; the source is
parsed, never imported: flexynesis.main sets torch's matmul precision to "medium" at import, main.py:24), and install() replaces
exactly the two methods + the model classes."""
import ast
import inspect
import os
import types

import pytest

REF_MAIN = "/root/reference/flexynesis/main.py"


def _self_attrs(fn_src, name="self"):
    tree = ast.parse(fn_src)
    return {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == name}


def test_adapter_signatures_are_the_references():
    from flexynesis_amd import adapters
    assert list(inspect.signature(adapters.objective).parameters) == ["self", "params", "current_step", "total_steps", "full_train"]
    assert inspect.signature(adapters.objective).parameters["full_train"].default is False
    assert list(inspect.signature(adapters.run_experiments).parameters) == ["self"]


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference exists in the build container only")
def test_adapters_read_only_what_the_reference_classes_set():
    import textwrap
    from flexynesis_amd import adapters
    tree = ast.parse(open(REF_MAIN).read())
    classes = {n.name: n for n in tree.body if isinstance(n, ast.ClassDef)}

    def method(cls, name):
        return next(f for f in classes[cls].body if isinstance(f, ast.FunctionDef) and f.name == name)

    def assigned_in_init(cls):
        init = method(cls, "__init__")
        return {t.attr for n in ast.walk(init) if isinstance(n, ast.Assign) for t in n.targets
                if isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id == "self"}
    # the reference's own signatures
    ref_obj = method("HyperparameterTuning", "objective")
    assert [a.arg for a in ref_obj.args.args] == ["self", "params", "current_step", "total_steps", "full_train"]
    assert [a.arg for a in method("FineTuner", "run_experiments").args.args] == ["self"]
    # every attribute the adapters READ is one the reference's __init__ sets
    hp_set, ft_set = assigned_in_init("HyperparameterTuning"), assigned_in_init("FineTuner")
    obj_reads = _self_attrs(textwrap.dedent(inspect.getsource(adapters.objective)))
    assert obj_reads <= hp_set, obj_reads - hp_set
    # ... and the adapter reads everything of the configuration the reference's objective reads (loader / progress-bar plumbing aside)
    ref_reads = {n.attr for n in ast.walk(ref_obj) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "self"}
    plumbing = {"DataLoader", "loader_dataset", "num_workers", "setup_trainer"}
    assert ref_reads - plumbing <= obj_reads, (ref_reads - plumbing) - obj_reads
    ft_reads = _self_attrs(textwrap.dedent(inspect.getsource(adapters.run_experiments)))
    ft_writes = {"val_loss_results", "best_config", "learning_rate", "model"}
    assert ft_reads - ft_writes <= ft_set, (ft_reads - ft_writes) - ft_set
    assert {"original_model", "dataset", "n_splits", "batch_size", "learning_rates", "max_epoch", "freeze_configs"} <= ft_reads


def test_install_replaces_the_two_methods_and_the_model_classes():
    from flexynesis_amd import adapters, models
    fake = types.ModuleType("fake_flexynesis_main")

    class HyperparameterTuning:
        def objective(self, params, current_step, total_steps, full_train=False):
            raise AssertionError("the reference's loop")

    class FineTuner:
        def run_experiments(self):
            raise AssertionError("the reference's loop")

    fake.HyperparameterTuning, fake.FineTuner = HyperparameterTuning, FineTuner
    fake.DirectPred = fake.supervised_vae = object
    done = adapters.install(fake)
    assert fake.HyperparameterTuning.objective is adapters.objective and fake.FineTuner.run_experiments is adapters.run_experiments
    assert fake.DirectPred is models.DirectPred and fake.supervised_vae is models.supervised_vae and not hasattr(fake, "GNN")
    assert set(done) == {"HyperparameterTuning.objective", "FineTuner.run_experiments", "DirectPred", "supervised_vae"}
    # a reference class maps to the engine's by name; an engine class passes through; an unknown one is refused
    ref_like = type("DirectPred", (), {})
    assert adapters.engine_class(ref_like) is models.DirectPred and adapters.engine_class(models.GNN) is models.GNN
    with pytest.raises(TypeError):
        adapters.engine_class(type("NoSuchModel", (), {}))
