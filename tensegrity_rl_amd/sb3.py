"""Optional adapter: present TensegrityVecEnv as a genuine stable_baselines3 VecEnv subclass."""


def as_sb3_vecenv(venv):
    from stable_baselines3.common.vec_env.base_vec_env import VecEnv  # raises ImportError when SB3 is absent

    class _Adapter(VecEnv):
        def __init__(self, inner):
            self.inner = inner
            super().__init__(inner.num_envs, inner.observation_space, inner.action_space)

        def reset(self):
            obs = self.inner.reset()
            self.reset_infos = self.inner.reset_infos
            return obs

        def step_async(self, actions):
            self.inner.step_async(actions)

        def step_wait(self):
            return self.inner.step_wait()

        def close(self):
            self.inner.close()

        def get_attr(self, attr_name, indices=None):
            return self.inner.get_attr(attr_name, indices)

        def set_attr(self, attr_name, value, indices=None):
            return self.inner.set_attr(attr_name, value, indices)

        def env_method(self, method_name, *a, indices=None, **k):
            return self.inner.env_method(method_name, *a, indices=indices, **k)

        def env_is_wrapped(self, wrapper_class, indices=None):
            return self.inner.env_is_wrapped(wrapper_class, indices)

    return _Adapter(venv)
