"""CPU oracle for the SHMGAN generator+discriminator train_step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``shmgan_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the reported baseline.

PARITY UNPINNED: the reference computes everything inside TensorFlow 2.8 /
Keras 2.8 / tensorflow-addons 0.17.1 (un-vendored, not installed here, no
network), and it ships no tests, golden vectors or fixtures for this path.
The oracle therefore restates the *published* semantics of those TF ops
(SAME padding, Conv2DTranspose phases, tfa InstanceNormalization op chain,
tf.image.ssim, rgb<->yuv matrices, Keras Adam) following the reference call
sites cited per function, and is pinned only by
  * the structural known answers in the reference's committed Keras summaries
    (parameter counts / layer shapes), and
  * two independent restatements checked against each other:
    ``tf_ops_np`` (NumPy, explicit index arithmetic, hand-written backward) and
    ``step_torch`` (PyTorch-CPU autograd, float64).
"""
