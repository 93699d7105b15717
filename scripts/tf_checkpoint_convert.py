"""TF checkpoint <-> the .safetensors container of gansynth_amd/checkpoint.py (same variable names, shapes and layouts).
Needs TensorFlow on the machine that runs it (not available in the build image, so this script is NOT exercised by the tests):

    python scripts/tf_checkpoint_convert.py to-safetensors  <tf checkpoint prefix>  <out.safetensors>
    python scripts/tf_checkpoint_convert.py to-tf           <in.safetensors>        <tf checkpoint prefix>

Reference side: models.py:123-130 (tf.train.Saver of the whole graph: trainable variables, `<var>/Adam`, `<var>/Adam_1`,
beta1_power[_1], beta2_power[_1], global_step)."""
import sys

import numpy as np


def to_safetensors(prefix, out):
    import tensorflow as tf
    import torch
    from safetensors.torch import save_file
    reader = tf.train.load_checkpoint(prefix)
    state = {}
    for name in reader.get_variable_to_shape_map():
        state[name] = torch.from_numpy(np.ascontiguousarray(reader.get_tensor(name)))
    save_file(state, out)
    print(f"{len(state)} tensors -> {out}")


def to_tf(path, prefix):
    import tensorflow as tf
    from safetensors.numpy import load_file
    state = {k: v for k, v in load_file(path).items() if not k.startswith("optimizer_steps")}
    tf1 = tf.compat.v1
    with tf1.Graph().as_default(), tf1.Session() as sess:
        variables = [tf1.get_variable(k, initializer=v) for k, v in state.items()]
        sess.run(tf1.global_variables_initializer())
        tf1.train.Saver(variables).save(sess, prefix, write_meta_graph=False)
    print(f"{len(state)} tensors -> {prefix}")


if __name__ == "__main__":
    if len(sys.argv) != 4 or sys.argv[1] not in ("to-safetensors", "to-tf"):
        raise SystemExit(__doc__)
    (to_safetensors if sys.argv[1] == "to-safetensors" else to_tf)(sys.argv[2], sys.argv[3])
