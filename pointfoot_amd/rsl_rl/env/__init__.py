from .vec_env import VecEnv
